"""Drop-in import path of the reference (``from model.loss import DistillationLoss`` ...): thin re-exports of deltakd_amd."""
