#!/bin/bash
# round 5: same-box A/B of bench.py settings given as NAME:ENV=VAL,ENV=VAL ... ; results in gpurun_out/$OUT/ab.txt
OUT=gpurun_out/${OUT:-r5ab}
mkdir -p $OUT
STEPS=${STEPS:-30}
for spec in "$@"; do
  name=${spec%%:*}
  envs=${spec#*:}
  [ "$envs" = "$spec" ] && envs=""
  for rep in 1 2; do
    line=$(env ${envs//,/ } timeout -k 10 300 python bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --traffic file ${BENCH_ARGS} 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), round(d['steady_ms_per_step'],3), d['config'].get('lrkd_mode'))")
    echo "$name rep$rep $line" | tee -a $OUT/ab.txt
  done
done
