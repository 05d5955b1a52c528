#!/bin/bash
# Same calling convention as the reference's exp/wasskd-deit-tiny.sh: GPU_IDS (e.g. 0,1,2,3) MASTER_PORT (e.g. 29501).
# One process per MI355X; gradients are averaged over RCCL/xGMI.  TEACHER defaults to the reference script's teacher.
# Flags as in the reference script (--alpha 0.5 is passed there too; the wasskd branch ignores it: model/loss.py:226).
# WASSKD_TYPE defaults to the reference's "sinkhorn", which needs the third-party geomloss package (absent, parity unpinned):
# tools/train.py stops with a message that says so.  WASSKD_TYPE=l1 runs the sorted-L1 variant (model/loss.py:187-199).
if [[ $# -ne 2 ]]; then echo "Usage: $0 GPU_IDS (example: 0,1,2,3) MASTER_PORT (example: 29501)"; exit 1; fi
GPU_IDS=$1; MASTER_PORT=$2
NUM_GPUS=$(echo $GPU_IDS | tr ',' '\n' | wc -l)
TEACHER=${TEACHER:-deit_small_distilled_patch16_224}
export HSA_ENABLE_IPC_MODE_LEGACY=0
HIP_VISIBLE_DEVICES=$GPU_IDS python -m torch.distributed.run --nnodes=1 --nproc-per-node $NUM_GPUS --master-addr 127.0.0.1 --master-port $MASTER_PORT tools/train.py \
    --student-model deit_tiny_patch16_224 --teacher-model $TEACHER --dataset cifar-100 --epochs ${EPOCHS:-300} --batch-size 256 \
    --lr 5e-4 --weight-decay 1e-4 --gpus $GPU_IDS --alpha 0.5 --distillation-type wasskd --wasskd-type ${WASSKD_TYPE:-sinkhorn} \
    --log-file logs/wasskd-deit-tiny-cifar100.log --save-dir checkpoints/wasskd-deit-tiny-cifar100
