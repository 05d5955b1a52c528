"""CPU oracle for the DeiT distillation step.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  Nothing under ``deltakd_amd/`` (the product path)
imports it; the product path raises when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ``oracle/loss_ref.py`` (DistillationLoss branches, random_masking, aux
    modules) is PINNED: ``oracle/gen_golden.py`` imports the reference's own
    ``model/loss.py``, ``model/models.py`` and ``model/misc.py`` in the build
    container and the restatement reproduces their losses and gradients
    (``tests/golden/*.npz``).
  * ``oracle/vit_ref.py`` restates timm==0.9.12's VisionTransformer /
    VisionTransformerDistilled, a third-party dependency that is absent from
    /root/reference and from this image -> ViT arithmetic is "parity unpinned"
    against timm itself; it is cross-checked against HF ``transformers``'
    DeiT implementation built from a local config (``gen_golden.py``).
"""
