#!/bin/bash
set -o pipefail
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_attn192_gpu.py -x -q -m gpu > $O/t_attn192.log 2>&1; rc=$?; echo "attn192 tests rc=$rc"; tail -5 $O/t_attn192.log
[ $rc -ne 0 ] && { grep -E "Error|assert|rel " $O/t_attn192.log | head -20; exit 1; }
timeout -k 10 120 python tools_dev/attn192_bwd_bench.py > $O/attn192_bwd_bench.txt 2>&1; echo "bench rc=$?"; tail -2 $O/attn192_bwd_bench.txt
timeout -k 10 600 python -m pytest tests/test_mlp192_gpu.py tests/test_parity_gpu.py tests/test_engine_gpu.py -x -q -m gpu > $O/t_blocks.log 2>&1; echo "block-level tests rc=$?"; tail -3 $O/t_blocks.log
run() {  # name, config, env...
  name=$1; cfg=$2; shift; shift
  env "$@" python bench.py --config $cfg --steps 30 --warmup 6 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; rc=$?
  python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    rs=j["roofline_student"]
    print("$name rc=$rc", round(j["value"]), round(j["ms_per_step"],3), round(j["steady_ms_per_step"],3), "student frac", round(rs["mfma"]["frac"],4), "bwd ms", round(rs["ms"],3), "fwd ms", round(rs["student_block_fwd"]["ms"],3))
except Exception as e:
    print("$name rc=$rc FAILED", e)
PY
}
run none_fused none A=1
run none_noln none DKD_ATTN_BWD_NO_LN=1
run none_nofuse none DKD_NO_ATTN_BWD_FUSION=1
run lrkd_look2_w2s4 lrkd DKD_LOOKAHEAD=2 DKD_LRKD_WARM_ITERS=2 DKD_LRKD_RITZ_SWEEPS=4
run lrkd_look2_w3s4 lrkd DKD_LOOKAHEAD=2 DKD_LRKD_WARM_ITERS=3 DKD_LRKD_RITZ_SWEEPS=4
run lrkd_look2_w1s2 lrkd DKD_LOOKAHEAD=2
run lrkd_look3_w1s2 lrkd DKD_LOOKAHEAD=3
