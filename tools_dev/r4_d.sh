#!/bin/bash
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 400 python tools_dev/lowrank_shift_probe.py 14 > $O/lowrank_shift_probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/lowrank_shift_probe.txt | tail -120
timeout -k 10 300 python tools_dev/aten_glue_trace.py none 3 > $O/aten_glue_none.txt 2>&1; echo "glue none rc=$?"; tail -40 $O/aten_glue_none.txt
timeout -k 10 300 python tools_dev/aten_glue_trace.py lrkd 3 > $O/aten_glue_lrkd.txt 2>&1; echo "glue lrkd rc=$?"; tail -45 $O/aten_glue_lrkd.txt
