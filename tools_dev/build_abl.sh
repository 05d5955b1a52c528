#!/bin/bash
# Builds ablation variants of libdkd.so (only mlp192.hip differs): tools_dev/bin/libdkd_abl<N>.so for each N given.
# Use with DKD_LIB=tools_dev/bin/libdkd_abl<N>.so python tools_dev/mlp192_bench.py
set -e
cd "$(dirname "$0")/.."
mkdir -p tools_dev/bin
OBJ=deltakd_amd/lib/obj
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DDKD_MLP_ABL=$n -c deltakd_amd/csrc/mlp192.hip -o tools_dev/bin/mlp192_abl$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/api.o $OBJ/gemm.o $OBJ/attn.o $OBJ/norm.o $OBJ/elementwise.o $OBJ/loss.o $OBJ/loss_ext.o $OBJ/lowrank.o tools_dev/bin/mlp192_abl$n.o $OBJ/block.o -o tools_dev/bin/libdkd_abl$n.so &
done
wait
ls -la tools_dev/bin/*.so
