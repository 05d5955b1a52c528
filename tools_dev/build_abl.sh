#!/bin/bash
# Builds ablation variants of libdkd.so: tools_dev/bin/libdkd_abl<N>.so for each N given (-DDKD_MLP_ABL=N in mlp192.hip, or, with
# FILE=attn192 MACRO=DKD_ATTN192_ABL in the environment, that file / macro).
# Use with DKD_LIB=tools_dev/bin/libdkd_abl<N>.so python tools_dev/mlp192_bench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools_dev/bin
OBJ=deltakd_amd/lib/obj
FILE=${FILE:-mlp192}
MACRO=${MACRO:-DKD_MLP_ABL}
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -D$MACRO=$n -c deltakd_amd/csrc/$FILE.hip -o tools_dev/bin/${FILE}_abl$n.o &
done
wait
for n in "$@"; do
  others=$(ls $OBJ/*.o | grep -v "/$FILE.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others tools_dev/bin/${FILE}_abl$n.o -o tools_dev/bin/libdkd_abl$n.so &
done
wait
ls -la tools_dev/bin/*.so
