"""Runtime helpers with the names of /root/reference/tools/utils.py that the step loop needs (:23-88): distributed init
over RCCL ("nccl" backend on ROCm), device selection, seeding.  Checkpoint / finetune helpers are out of scope."""
import datetime
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def setup_distributed(args):
    """tools/utils.py:23-65: env:// rendezvous from torchrun; one process per GPU."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
        args.distributed = args.world_size > 1
    else:
        args.rank, args.world_size, args.gpu, args.distributed = 0, 1, 0, False
    if args.distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(args.gpu)
        dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"), world_size=args.world_size,
                                rank=args.rank, timeout=datetime.timedelta(seconds=1800))
        dist.barrier()
    return args


def setup_device(args):
    if getattr(args, "device", None):
        return torch.device(args.device)
    if torch.cuda.is_available():
        return torch.device("cuda", getattr(args, "gpu", 0))
    raise RuntimeError("deltakd_amd needs an MI355X (no CPU path); pass --device explicitly to override")


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_model_state(model):
    return (model.module if hasattr(model, "module") else model).state_dict()


def remove_module_prefix(state_dict):
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in state_dict.items()}
