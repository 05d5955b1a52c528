"""Same import path as /root/reference/model/loss.py; implementation: deltakd_amd.losses (HIP kernels behind libdkd.so)."""
from deltakd_amd.losses import (DistillationLoss, LabelSmoothingCrossEntropy, LowRankTargets, SoftTargetCrossEntropy,  # noqa: F401
                                call_base_loss, lrkd_loss, lrkd_targets)
from deltakd_amd.losses_ext import curkd_loss, diffkd_loss, mgd_loss, saliency_mgd_loss, vitkd_loss, wasskd_l1_loss  # noqa: F401
