#!/bin/bash
# Same calling convention as the reference's exp/saliencymgd-deit-tiny.sh: GPU_IDS (e.g. 0,1,2,3) MASTER_PORT (e.g. 29501).
if [[ $# -ne 2 ]]; then echo "Usage: $0 GPU_IDS (example: 0,1,2,3) MASTER_PORT (example: 29501)"; exit 1; fi
GPU_IDS=$1; MASTER_PORT=$2
NUM_GPUS=$(echo $GPU_IDS | tr ',' '\n' | wc -l)
TEACHER=${TEACHER:-deit_small_distilled_patch16_224}
export HSA_ENABLE_IPC_MODE_LEGACY=0
HIP_VISIBLE_DEVICES=$GPU_IDS python -m torch.distributed.run --nnodes=1 --nproc-per-node $NUM_GPUS --master-addr 127.0.0.1 --master-port $MASTER_PORT tools/train.py \
    --student-model deit_tiny_patch16_224 --teacher-model $TEACHER --dataset cifar-100 --epochs ${EPOCHS:-300} --batch-size 256 \
    --lr 5e-4 --weight-decay 1e-4 --gpus $GPU_IDS --distillation-type saliency_mgd --saliency-method 1 --saliency-mask-ratio 0.5 \
    --log-file logs/saliencymgd-deit-tiny-cifar100.log --save-dir checkpoints/saliencymgd-deit-tiny-cifar100
