"""Source-domain pre-training side of DG-TTA on the GPU: the trainers' forward pre-hooks
(dg_tta/pretraining/nnUNetTrainer_{GIN,MIND,GIN_MIND}.py:38-59) and the discrete low-resolution augmentation of the
MultiRes trainers (dg_tta/pretraining/discrete_downsampling.py).  nnU-Net's training loop itself stays with nnU-Net."""
from .discrete_downsampling import SimulateDiscreteLowResolutionTransform, augment_discrete_linear_downsampling  # noqa: F401
from .hooks import build_network_architecture, register_dg_hooks  # noqa: F401
