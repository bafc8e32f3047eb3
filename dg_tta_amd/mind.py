"""MIND3D descriptor — drop-in for dg_tta/mind.py (MIND3D :97-164, mind_hook :167-168), computed by the fused HIP
kernels of csrc/mind3d.hip.  The Gaussian noise of mind.py:150 is drawn with torch.randn on the input's device
(same generator the reference uses) and handed to the kernel, so seeding behaves exactly as in the reference."""
import torch

from . import ops


class MIND3D(torch.nn.Module):
    def __init__(self, delta=1, sigma=1, randn_weighting=0.05) -> None:
        super().__init__()
        if delta != 1 or sigma != 1:
            raise NotImplementedError("the HIP MIND3D kernel is built for delta=1, sigma=1 (all the reference ever uses)")
        self.delta, self.sigma, self.randn_weighting = delta, sigma, randn_weighting
        self.out_channels = 12

    def forward(self, img, noise=None, out_dtype=torch.float32):
        """Returns a logical [B,12,D,H,W] tensor.  Its memory is voxel-major with rows padded to 16 channels
        ([B,D,H,W,16], channels 12..15 zero), i.e. exactly what HipPlainConvUNet's first conv reads."""
        b, _, d, h, w = img.shape
        if noise is None:
            noise = torch.randn((b, 12, d, h, w), dtype=torch.float32, device=img.device)
        buf = ops.mind3d(img, noise, self.randn_weighting, out_format="ndhwc", out_ldc=16, out_dtype=out_dtype)
        return buf[..., :12].permute(0, 4, 1, 2, 3)


def mind_hook(module, input):
    return MIND3D().forward(*input, out_dtype=getattr(module, "act_dtype", torch.float32))
