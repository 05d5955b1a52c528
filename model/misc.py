"""Same import path as /root/reference/model/misc.py; implementation: deltakd_amd.misc."""
from deltakd_amd.misc import masking_indices, random_masking, saliency_masking, saliency_scores  # noqa: F401
