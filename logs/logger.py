"""Same import path as /root/reference/logs/logger.py (MetricLogger / SmoothedValue); implementation: deltakd_amd.logger."""
import datetime
import logging
import sys

import torch.distributed as dist

from deltakd_amd.logger import MetricLogger, SmoothedValue  # noqa: F401


def setup_logger(log_file):
    """logs/logger.py:10-24: file + stdout handlers on rank 0."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    logger = logging.getLogger(__name__)
    logger.setLevel(logging.INFO)
    fmt = logging.Formatter('%(asctime)s - %(levelname)s - %(message)s')
    if rank == 0:
        for h in (logging.FileHandler(log_file), logging.StreamHandler(sys.stdout)):
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger


def get_timestamped_log_file_path(log_file):
    return f"{log_file}_{datetime.datetime.now().strftime('%Y%m%d_%H%M%S')}"
