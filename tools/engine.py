"""Same import path as /root/reference/tools/engine.py; implementation: deltakd_amd.engine."""
from deltakd_amd.engine import train_one_epoch, validate  # noqa: F401
