#!/bin/bash
# round 5: the whole GPU suite, then the default bench line (what the driver runs)
OUT=gpurun_out/${1:-r5full}
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/gpu_suite.log 2>&1
echo "gpu suite rc=$?" | tee $OUT/summary.txt
tail -4 $OUT/gpu_suite.log | tee -a $OUT/summary.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err
echo "bench rc=$?" | tee -a $OUT/summary.txt
grep '^{' $OUT/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', round(d['value']), 'ms', round(d['ms_per_step'], 3), 'steady', round(d['steady_ms_per_step'], 3))
print('lrkd_mode', d['config'].get('lrkd_mode'))
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['avg_launch_us'])
print('student', d['roofline_student']['mfma']['frac'], d['roofline_student']['ms'])
print('others', json.dumps(d.get('other_configs')))
print('cpu', d.get('cpu_baseline', {}).get('value'))
" | tee -a $OUT/summary.txt
