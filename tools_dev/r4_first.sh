#!/bin/bash
# round 4, first GPU call: the new tests, RCCL smoke, bench self-launch rehearsal, a baseline bench line on this box
set -o pipefail
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "rccl or two_ranks" --durations=5 > $O/t_dp.log 2>&1; echo "dp tests rc=$?"; tail -3 $O/t_dp.log
python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "layernorm_fold or folded" -s --durations=5 > $O/t_fold.log 2>&1; echo "fold tests rc=$?"; tail -3 $O/t_fold.log
python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "topk or fused_adamw" > $O/t_misc.log 2>&1; echo "misc rc=$?"; tail -3 $O/t_misc.log
python -m pytest tests/test_loss_curve_gpu.py -x -q -m gpu -s --durations=5 > $O/t_curve.log 2>&1; echo "curve rc=$?"; tail -8 $O/t_curve.log
python tools_dev/rccl_smoke.py > $O/rccl_smoke.txt 2>&1; echo "rccl smoke rc=$?"; tail -2 $O/rccl_smoke.txt
python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2_onegpu.txt 2>&1; echo "bench --gpus 2 rc=$? (expected 2)"; tail -2 $O/bench_gpus2_onegpu.txt
DKD_DIST_BACKEND=gloo DKD_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_selflaunch.txt 2>&1; echo "self-launch rehearsal rc=$?"; tail -c 600 $O/bench_dp2_gloo_selflaunch.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lrkd.json 2> $O/bench_lrkd.err; echo "bench rc=$?"; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4a/bench_lrkd.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline_student"]["mfma"]["frac"], j["roofline_student"]["ms"])
PY
python bench.py --config none --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_none.json 2> $O/bench_none.err; echo "bench none rc=$?"; python - <<'PY'
import json
j=json.loads(open("gpurun_out/r4a/bench_none.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline_student"]["mfma"]["frac"], j["roofline_student"]["ms"], j["roofline_student"]["student_block_fwd"])
PY
