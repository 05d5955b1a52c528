"""Runtime helpers with the names of /root/reference/tools/utils.py that the step loop needs (:23-88): distributed init
over RCCL ("nccl" backend on ROCm), device selection, seeding, and the checkpoint helpers (:90-160).

Checkpoint wire format = the reference's: one ``torch.save``d dict ``{epoch, model, optimizer, scheduler,
scaler}``; ``model`` has timm's key names (deltakd_amd.vit keeps them), ``optimizer`` torch.optim.AdamW's layout
(deltakd_amd.optim.FusedAdamW reads and writes it), so a file written by either side resumes on the other."""
import datetime
import math
import os
import random
import shutil

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


def save_checkpoint(state, is_best, filename):
    """tools/utils.py:90-93: write the dict; a best-so-far copy goes next to it with 'pth' -> 'best.pth' in the name."""
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, filename.replace('pth', 'best.pth'))


def _read(filename):
    if not os.path.exists(filename):
        raise FileNotFoundError(f"Checkpoint file not found: {filename}")
    # tensors, numbers, strings and containers only (what either side writes): nothing in the file is executed
    return torch.load(filename, map_location='cpu', weights_only=True)


def load_checkpoint(model, optimizer, scheduler, scaler, filename):
    """tools/utils.py:96-103: restore everything for --resume; returns (epoch, model, optimizer, scheduler, scaler)."""
    ckpt = _read(filename)
    (model.module if hasattr(model, "module") else model).load_state_dict(remove_module_prefix(ckpt['model']))
    optimizer.load_state_dict(ckpt['optimizer'])
    scheduler.load_state_dict(ckpt['scheduler'])
    scaler.load_state_dict(ckpt['scaler'])
    return ckpt['epoch'], model, optimizer, scheduler, scaler


def load_model(model, filename):
    """tools/utils.py:106-109: weights only."""
    (model.module if hasattr(model, "module") else model).load_state_dict(remove_module_prefix(_read(filename)['model']))
    return model


def enable_finetune_mode(model, model_ckpt):
    """tools/utils.py:112-160: load a pretrained state dict into a model with another head size and/or patch grid.
    Head weights of a different shape are dropped; the patch part of ``pos_embed`` is resized bicubically (align_corners False)
    to the model's grid, the prefix-token part is taken from the checkpoint (or, when the checkpoint stores patch positions only,
    from the model); everything else loads non-strictly."""
    own = model.state_dict()
    for k in ('head.weight', 'head.bias'):
        if k in model_ckpt and k in own and model_ckpt[k].shape != own[k].shape:
            print(f"Removing key {k} from pretrained checkpoint")
            del model_ckpt[k]
    pos = model_ckpt['pos_embed']
    n_patches = model.patch_embed.num_patches
    n_extra = model.pos_embed.shape[1] - n_patches
    if pos.shape[1] == n_patches:
        extra, grid = model.pos_embed[:, :n_extra].detach().to(pos.dtype).cpu(), pos
    elif pos.shape[1] == n_extra + n_patches:
        extra, grid = pos[:, :n_extra], pos[:, n_extra:]
    else:
        print(f"Warning: Checkpoint pos_embed token count ({pos.shape[1]}) does not match expected ({n_extra + n_patches}). "
              "Adjusting token selection.")
        extra, grid = pos[:, :n_extra], pos[:, n_extra:n_extra + n_patches]
    old, new = int(math.sqrt(grid.shape[1])), int(math.sqrt(n_patches))
    grid = grid.reshape(-1, old, old, pos.shape[-1]).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(new, new), mode='bicubic', align_corners=False)
    model_ckpt['pos_embed'] = torch.cat((extra, grid.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)
    model.load_state_dict(model_ckpt, strict=False)
