#!/bin/bash
# LRKD tracker accuracy settings vs step time (chain on its own stream / on the teacher stream)
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; rc=$?
  python - <<PY
import json
try:
    j=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name rc=$rc", round(j["value"]), round(j["ms_per_step"],3), round(j["steady_ms_per_step"],3))
except Exception as e:
    print("$name rc=$rc FAILED", e)
PY
}
run w1s2 DKD_LRKD_WARM_ITERS=1 DKD_LRKD_RITZ_SWEEPS=2
run w2s4 DKD_LRKD_WARM_ITERS=2 DKD_LRKD_RITZ_SWEEPS=4
run w3s4 DKD_LRKD_WARM_ITERS=3 DKD_LRKD_RITZ_SWEEPS=4
run w4s6 DKD_LRKD_WARM_ITERS=4 DKD_LRKD_RITZ_SWEEPS=6
run w8s12 DKD_LRKD_EXACT=1
run w1s2_same DKD_LRKD_WARM_ITERS=1 DKD_LRKD_RITZ_SWEEPS=2 DKD_LRKD_STREAM=0
run w4s6_same DKD_LRKD_WARM_ITERS=4 DKD_LRKD_RITZ_SWEEPS=6 DKD_LRKD_STREAM=0
run w8s12_same DKD_LRKD_EXACT=1 DKD_LRKD_STREAM=0
