#!/bin/bash
set -o pipefail
O=gpurun_out/r4f; mkdir -p $O
PROBE_SETTINGS="1,3;1,4;1,6;1,12" timeout -k 10 400 python tools_dev/lowrank_shift_probe.py 14 > $O/lowrank_shift_probe_sweeps.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/lowrank_shift_probe_sweeps.txt | grep -E "warm_iters|call  7|call 12|call  1:" 
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
run w1s4 DKD_LRKD_WARM_ITERS=1 DKD_LRKD_RITZ_SWEEPS=4
run w1s6 DKD_LRKD_WARM_ITERS=1 DKD_LRKD_RITZ_SWEEPS=6
run w4s6_look2 DKD_LRKD_WARM_ITERS=4 DKD_LRKD_RITZ_SWEEPS=6 DKD_LOOKAHEAD=2
run w8s12_look2 DKD_LRKD_EXACT=1 DKD_LOOKAHEAD=2
run w1s2_look2 DKD_LOOKAHEAD=2
