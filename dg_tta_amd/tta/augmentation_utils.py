"""Spatial augmentation parameters — mirrors dg_tta/tta/augmentation_utils.py:156-170 (get_rand_affine).
The deformable path of the reference (get_disp_field, :138-153) raises TypeError there (it passes an argument
get_rf_field does not accept), so there is nothing to be compatible with; it is reported as unsupported here."""
import torch

from ..utils import cpu_generator


def get_rand_affine(batch_size, strength=0.05, flip=False):
    """theta = I(3x4) + strength*randn (CPU generator, like the reference) and its inverse; both [B,3,4] on the CPU."""
    top = torch.randn(batch_size, 3, 4, generator=cpu_generator()) * strength + torch.eye(3, 4).unsqueeze(0)
    bottom = torch.tensor([0, 0, 0, 1]).view(1, 1, 4).repeat(batch_size, 1, 1)
    affine = torch.cat((top, bottom), 1)
    if flip:
        signs = torch.cat([(2 * (torch.rand(3, generator=cpu_generator()) > 0.5).float() - 1), torch.tensor([1.0])])
        affine = affine @ torch.diag(signs)
    return affine[:, :3], affine.inverse()[:, :3]


def get_disp_field(*args, **kwargs):
    raise NotImplementedError("spatial_aug_type='deformable' is broken in the reference (TypeError at "
                              "augmentation_utils.py:141-148) and is not provided; use 'affine'")
