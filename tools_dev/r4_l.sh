#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
for i in 1 2; do for v in 0 1; do
  DKD_ATTN_BWD_NO_LN=$v python bench.py --config none --steps 40 --warmup 6 --no-cpu-baseline > $O/none_noln${v}_$i.json 2>/dev/null
  python -c "
import json; j=json.loads(open('$O/none_noln${v}_$i.json').read().strip().splitlines()[-1]); rs=j['roofline_student']; print('no_ln=$v run $i', round(j['value']), round(j['ms_per_step'],3), round(rs['mfma']['frac'],4), round(rs['ms'],3), round(rs['student_block_fwd']['ms'],3))"
done; done
( time python -m pytest tests -x -q -m gpu --durations=15 ) > $O/full_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -25 $O/full_gpu_suite.log
