"""Same import path as /root/reference/model/models.py; implementation: deltakd_amd.models."""
from deltakd_amd.models import (DATASET_NUM_CLASSES, DenoisingNetwork, Generation, SimpleAttention, SimpleCrossAttention,  # noqa: F401
                                attach_aux, forward_with_features, load_teacher_student_model)
from deltakd_amd.vit import REGISTRY, VisionTransformer, create_model  # noqa: F401
