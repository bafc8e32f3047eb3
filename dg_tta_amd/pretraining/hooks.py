"""What the DG trainers' `build_network_architecture` does around the network (reference:
dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:38-59, nnUNetTrainer_GIN.py:38-58, nnUNetTrainer_MIND.py:37-56): the input
channel count of the descriptor (12 for MIND), internal augmentation switched on, and the forward pre-hooks registered
in the reference's order (gin_hook first, then mind_hook).  The hooks run the HIP kernels (csrc/gin.hip, csrc/mind3d.hip)
for any batch size, GIN with one random kernel chain per batch item (groups = nb, gin.py:168-230)."""
from ..gin import gin_hook
from ..mind import mind_hook
from ..unet import PLANS_3D_FULLRES, HipPlainConvUNet
from ..utils import enable_internal_augmentation


def trainer_spec(trainer_name):
    """(network input channels, [pre-hooks]) of a DG trainer name (with or without the _MultiRes suffix)."""
    name = trainer_name.replace("_MultiRes", "")
    if name.endswith("GIN_MIND"):
        return 12, [gin_hook, mind_hook]
    if name.endswith("MIND"):
        return 12, [mind_hook]
    if name.endswith("GIN"):
        return 1, [gin_hook]
    raise NotImplementedError(f"trainer {trainer_name}: only the DG-TTA trainers (GIN / MIND / GIN_MIND) are built")


def register_dg_hooks(network, trainer_name="nnUNetTrainer_GIN_MIND"):
    """enable_internal_augmentation() + register_forward_pre_hook(gin_hook / mind_hook), as the trainers do."""
    _, hooks = trainer_spec(trainer_name)
    enable_internal_augmentation()
    return [network.register_forward_pre_hook(h) for h in hooks]


def build_network_architecture(cfg=None, trainer_name="nnUNetTrainer_GIN_MIND", act_dtype=None, num_classes=None):
    """The network a DG trainer trains: PlainConvUNet topology from the plans (default: the 3d_fullres plans of the
    reference's model), `num_input_channels` overridden by the trainer, hooks registered."""
    import torch
    cfg = dict(PLANS_3D_FULLRES if cfg is None else cfg)
    cfg["in_channels"], _ = trainer_spec(trainer_name)
    if num_classes is not None:
        cfg["num_classes"] = num_classes
    net = HipPlainConvUNet(cfg, act_dtype=torch.float32 if act_dtype is None else act_dtype)
    register_dg_hooks(net, trainer_name)
    return net
