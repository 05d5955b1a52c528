#!/bin/bash
# round 4, third GPU call: ablation timings of the fused attention backward, wide-kernel loop-only rate at K = 3072, ATen glue call sites,
# the new tracker tests
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
for n in 0 2 4 6 8 16 22; do
  lib=deltakd_amd/lib/libdkd.so; [ $n -ne 0 ] && lib=tools_dev/bin/libdkd_abl$n.so
  echo "abl $n: $(DKD_LIB=$lib timeout -k 10 120 python tools_dev/attn192_bwd_bench.py 2>&1 | tail -1)" | tee -a $O/attn192_bwd_ablations.txt
done
echo "--- wide kernel, whole (K scan)" | tee $O/gemm_kscan.txt
timeout -k 10 120 python tools_dev/gemm_kscan.py 2>&1 | tail -4 | tee -a $O/gemm_kscan.txt
echo "--- wide kernel, no epilogue (DKD_NT256_ABL=1)" | tee -a $O/gemm_kscan.txt
DKD_LIB=tools_dev/bin/libdkd_gemm_abl1.so DKD_NT256_ABL=1 timeout -k 10 120 python tools_dev/gemm_kscan.py 2>&1 | tail -6 | tee -a $O/gemm_kscan.txt
timeout -k 10 300 python tools_dev/aten_glue_trace.py none 3 > $O/aten_glue_none.txt 2>&1; echo "glue none rc=$?"; tail -40 $O/aten_glue_none.txt
timeout -k 10 300 python tools_dev/aten_glue_trace.py lrkd 3 > $O/aten_glue_lrkd.txt 2>&1; echo "glue lrkd rc=$?"; tail -45 $O/aten_glue_lrkd.txt
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s -k "shifting or exact_mode or headline" --durations=5 > $O/t_tracker.log 2>&1; echo "tracker tests rc=$?"; grep -E "call|worst|passed|failed|Error|assert" $O/t_tracker.log | tail -60
