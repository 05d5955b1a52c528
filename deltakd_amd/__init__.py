"""deltakd_amd: MI355X-native (gfx950) DeiT distillation training step.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) over the C ABI of libdkd.so
(include/dkd.h): hand-written HIP kernels.  There is no CPU / eager fallback.
"""
__version__ = "0.1.0"
