#!/bin/bash
O=gpurun_out/r4t; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_attn192_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools_dev/attn192_bench.py 2>&1 | tail -2 | tee $O/attn192_fwd_bench.txt
for i in 1 2; do for v in default noproj; do
  case $v in default) e="A=1";; noproj) e="DKD_NO_ATTN_FWD_PROJ=1";; esac
  env $e python bench.py --config none --steps 40 --warmup 6 --no-cpu-baseline --traffic file > $O/none_${v}_$i.json 2>/dev/null
  python -c "
import json; j=json.loads([l for l in open('$O/none_${v}_$i.json') if l.startswith('{')][-1]); rs=j['roofline_student']; print('$v run $i', round(j['value']), round(j['ms_per_step'],3), round(rs['mfma']['frac'],4), round(rs['ms'],3), round(rs['student_block_fwd']['ms'],3))"
done; done | tee $O/none_ab.txt
( time python -m pytest tests -q -m gpu ) > $O/full_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -6 $O/full_gpu_suite.log
