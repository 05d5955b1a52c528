#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
python tools_dev/attn192_bwd_bench.py 2>&1 | tail -1 | tee $O/attn192_bwd_bench.txt
for i in 1 2; do for v in default ln nofuse; do
  case $v in default) e="A=1";; ln) e="DKD_ATTN_BWD_LN=1";; nofuse) e="DKD_NO_ATTN_BWD_FUSION=1";; esac
  env $e python bench.py --config none --steps 40 --warmup 6 --no-cpu-baseline > $O/none_${v}_$i.json 2>/dev/null
  python -c "
import json; j=json.loads(open('$O/none_${v}_$i.json').read().strip().splitlines()[-1]); rs=j['roofline_student']; print('$v run $i', round(j['value']), round(j['ms_per_step'],3), round(rs['mfma']['frac'],4), round(rs['ms'],3), round(rs['student_block_fwd']['ms'],3))"
done; done | tee $O/none_ab.txt
( time python -m pytest tests -q -m gpu --durations=12 ) > $O/full_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -22 $O/full_gpu_suite.log
