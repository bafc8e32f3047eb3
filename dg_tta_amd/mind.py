"""MIND3D descriptor — drop-in for dg_tta/mind.py (MIND3D :97-164, mind_hook :167-168), computed by the fused HIP
kernels of csrc/mind3d.hip.  The Gaussian noise of mind.py:150 is drawn with torch.randn on the input's device
(same generator the reference uses) and handed to the kernel, so seeding behaves exactly as in the reference."""
import torch

from . import ops
from ._state import state_of
from .utils import device_generator


class MIND3D(torch.nn.Module):
    def __init__(self, delta=1, sigma=1, randn_weighting=0.05) -> None:
        super().__init__()
        if delta not in (1, 2) or not (0 < sigma <= 2):
            raise NotImplementedError("the HIP MIND3D kernels are built for delta in {1, 2} and 0 < sigma <= 2 "
                                      "(the reference itself only ever uses delta=1, sigma=1)")
        self.delta, self.sigma, self.randn_weighting = delta, sigma, randn_weighting
        self.out_channels = 12

    def forward(self, img, noise=None, out_dtype=torch.float32, groups=1):
        """Returns a logical [B,12,D,H,W] tensor.  Its memory is voxel-major with rows padded to 16 channels
        ([B,D,H,W,16], channels 12..15 zero), i.e. exactly what HipPlainConvUNet's first conv reads."""
        b, _, d, h, w = img.shape
        if noise is None:
            noise = torch.randn((b, 12, d, h, w), dtype=torch.float32, device=img.device, generator=device_generator())
        buf = ops.mind3d(img, noise, self.randn_weighting, out_format="ndhwc", out_ldc=16, out_dtype=out_dtype, groups=groups,
                         delta=self.delta, sigma=self.sigma)
        return buf[..., :12].permute(0, 4, 1, 2, 3)


# Noise tensors drawn ahead of time by the batched two-branch path (tta.calc_both_branches): the reference draws a
# branch's MIND noise inside that branch's forward pass, i.e. BEFORE the other branch's GIN draws; pre-drawing keeps
# that order on the device generator when both branches run as one batch.  The hand-over is per MODEL (_state.py): two
# TTA instances in one process never take each other's noise or descriptor.
def draw_noise_(slot):
    """mind.py:150's torch.randn draw for one branch, written in place into its slot of the batched noise tensor (device
    generator, as the reference).  A function of its own so that parity tests can route it through the CPU generator."""
    return slot.normal_(generator=device_generator())


def hook_owner(model):
    """The module mind_hook is registered on (the model itself in every supported configuration)."""
    for m in model.modules():
        if any(h is mind_hook for h in m._forward_pre_hooks.values()):
            return m
    return model


def push_noise(model, noise, groups=1):
    state_of(hook_owner(model)).noise.append((noise, groups))


def push_features(model, feat):
    """Descriptor computed ahead of the network pass (tta.prepare_both_branches, on a side stream): the next mind_hook call
    of THIS model on an input of the same batch / spatial shape returns it instead of computing MIND again."""
    state_of(hook_owner(model)).features.append(feat)


def clear_noise(model):
    st = state_of(hook_owner(model))
    st.noise.clear()
    st.features.clear()


def uses_mind_hook(model):
    return any(h is mind_hook for m in model.modules() for h in m._forward_pre_hooks.values())


class mind_groups:
    """Context: `model`'s mind_hook treats the batch as `groups` independent calls (one variance-clamp mean per group)."""

    def __init__(self, model, groups):
        self.state, self.groups = state_of(hook_owner(model)), groups

    def __enter__(self):
        self.prev, self.state.forced_groups = self.state.forced_groups, self.groups

    def __exit__(self, *exc):
        self.state.forced_groups = self.prev


def mind_hook(module, input):
    if module is None:          # called as a plain function (the reference's hook ignores `module` as well)
        return MIND3D().forward(*input)
    st = state_of(module)
    if st.features:
        cand = st.features[0]
        if cand.shape[0] == input[0].shape[0] and tuple(cand.shape[2:]) == tuple(input[0].shape[2:]):
            return st.features.pop(0)
    noise, groups = None, 1
    if st.noise:
        cand = st.noise[0][0]
        if cand.shape[0] == input[0].shape[0] and tuple(cand.shape[2:]) == tuple(input[0].shape[2:]):
            noise, groups = st.noise.pop(0)
    if st.forced_groups is not None and noise is None and input[0].shape[0] % st.forced_groups == 0:
        groups = st.forced_groups
    return MIND3D().forward(*input, noise=noise, out_dtype=getattr(module, "act_dtype", torch.float32), groups=groups)
