"""GIN intensity augmentation — drop-in for dg_tta/gin.py (gin_aug :233-241, gin_hook :244-247).

Draw order and generators follow the reference exactly (gin.py:193-195 alpha on x.device; :65-66, :94-103 kernel size,
kernels and shifts on the CPU generator, then moved to the device); the convolution chain, blend and Frobenius
re-normalisation run in the fused HIP kernels of csrc/gin.hip."""
import torch

from . import ops
from .utils import cpu_generator, device_generator, get_internal_augmentation_enabled, upload_async

N_LAYER, INTERM_CHANNELS, SCALE_POOL = 4, 2, (1, 3)


def draw_gin_params(nb, device):
    alpha = torch.rand(nb, device=device, generator=device_generator())
    g = cpu_generator()
    chans = [1] + [INTERM_CHANNELS] * (N_LAYER - 1) + [1]
    ks, kers, shifts = [], [], []
    for cin, cout in zip(chans[:-1], chans[1:]):
        k = SCALE_POOL[int(torch.randint(high=len(SCALE_POOL), size=(1,), generator=g)[0])]
        kers.append(torch.randn([cout * nb, cin, k, k, k], generator=g))
        shifts.append(torch.randn([cout * nb, 1, 1, 1], generator=g) * 1.0)
        ks.append(k)
    up = upload_async(kers + shifts, device)          # one non-blocking copy for the whole chain
    return alpha, ks, up[:len(kers)], up[len(kers):]


def gin_aug(input):
    if input.dim() != 5 or input.shape[1] != 1:
        raise ValueError("gin_aug (HIP) expects [B,1,D,H,W]")
    alpha, ks, kers, shifts = draw_gin_params(input.shape[0], input.device)
    return ops.gin_chain(input, alpha, ks, kers, shifts)


def gin_hook(module, input):
    if get_internal_augmentation_enabled():
        return gin_aug(*input)
    return input
