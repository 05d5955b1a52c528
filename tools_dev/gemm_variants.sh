#!/bin/bash
# dev: compare GEMM pipeline variants on the teacher shapes (one process per variant: the knob is read once)
for v in 6; do
  echo "== variant $v"; DKD_GEMM_VARIANT=$v python tools_dev/gemm_bench.py t_qkv t_proj t_fc1 t_fc2 2>&1 | grep -E "t_|spot"
done
