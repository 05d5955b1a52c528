#!/bin/bash
# round 4, second GPU call: the fused attention-branch backward kernel (tests, timing), the lrkd-stream A/B, suite timing
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_attn192_gpu.py -x -q -m gpu > $O/t_attn192.log 2>&1; rc=$?; echo "attn192 tests rc=$rc"; tail -5 $O/t_attn192.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 120 python tools_dev/attn192_bwd_bench.py > $O/attn192_bwd_bench.txt 2>&1; echo "bench rc=$?"; tail -3 $O/attn192_bwd_bench.txt
timeout -k 10 600 python -m pytest tests/test_mlp192_gpu.py tests/test_parity_gpu.py tests/test_engine_gpu.py -x -q -m gpu --durations=8 > $O/t_blocks.log 2>&1; echo "block-level tests rc=$?"; tail -14 $O/t_blocks.log
for v in 1 0; do
  DKD_LRKD_STREAM=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_lrkd_stream$v.json 2> $O/bench_lrkd_stream$v.err; echo "bench lrkd stream=$v rc=$?"
  python - <<PY
import json
j=json.loads(open("$O/bench_lrkd_stream$v.json").read().strip().splitlines()[-1])
print("lrkd_stream=$v", j["value"], j["ms_per_step"], j["steady_ms_per_step"], j["roofline"]["frac"], j["roofline_student"]["mfma"]["frac"], j["roofline_student"]["ms"])
PY
done
for v in 0 1; do
  DKD_NO_ATTN_BWD_FUSION=$v python bench.py --config none --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_none_nofuse$v.json 2> $O/bench_none_nofuse$v.err; echo "bench none nofuse=$v rc=$?"
  python - <<PY
import json
j=json.loads(open("$O/bench_none_nofuse$v.json").read().strip().splitlines()[-1])
print("no_attn_bwd_fusion=$v", j["value"], j["ms_per_step"], j["roofline_student"]["mfma"]["frac"], j["roofline_student"]["ms"])
PY
done
