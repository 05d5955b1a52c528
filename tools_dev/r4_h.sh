#!/bin/bash
set -o pipefail
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 120 python tools_dev/attn192_bwd_debug.py 300 2>&1 | grep -v amdgpu | tee $O/debug300.txt
timeout -k 10 120 python tools_dev/attn192_bwd_debug.py 256 2>&1 | grep -v amdgpu | tee $O/debug256.txt
timeout -k 10 300 python -m pytest tests/test_attn192_gpu.py -q -m gpu > $O/t_attn192.log 2>&1; rc=$?; echo "attn192 tests rc=$rc"; tail -8 $O/t_attn192.log
