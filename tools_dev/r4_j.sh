#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
for n in 0 32 64 96 128; do
  lib=deltakd_amd/lib/libdkd.so; [ $n -ne 0 ] && lib=tools_dev/bin/libdkd_abl$n.so
  echo "abl $n: $(DKD_LIB=$lib timeout -k 10 120 python tools_dev/attn192_bwd_bench.py 2>&1 | tail -1)" | tee -a $O/attn192_bwd_phase_c_ablations.txt
done
