#!/bin/bash
# round 5: the LRKD chain -- kernel tests, accuracy / latency on shifting batches per setting, per-kernel durations (rocprofv3)
set -o pipefail
OUT=gpurun_out/${1:-r5a}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "lowrank or jacobi" > $OUT/kernel_tests.log 2>&1
echo "kernel tests rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/kernel_tests.log
PROBE_SETTINGS="${PROBE_SETTINGS:-1,2;4,6;8,12}" timeout -k 10 600 python tools_dev/lowrank_shift_probe.py 14 > $OUT/shift_probe.log 2>&1
echo "shift probe rc=$?" | tee -a $OUT/summary.txt
grep "ms/call\|spectrum" $OUT/shift_probe.log | tee -a $OUT/summary.txt
cd /tmp && PROBE_SETTINGS="8,12" timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o chain --output-format csv -- python3 $GRAFT_REPO_ROOT/tools_dev/lowrank_shift_probe.py 8 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
echo "rocprof rc=$?" | tee -a $GRAFT_REPO_ROOT/$OUT/summary.txt
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -i "lr_\|sgemm\|lowrank\|jacobi\|gemm_tn" $f | cut -c1-200 | tee -a $OUT/summary.txt
