"""Synthetic workload of SURVEY.md §8(d): a seeded MR-like volume with ellipsoid labels, a label mapping with
`C_opt` optimised classes, and seeded He-initialised nnUNet weights (the TS104 checkpoints cannot be downloaded)."""
import math

import torch


def synthetic_case(size=160, k=15, seed=20240704):
    """Returns data [1+K, S,S,S] fp32: smooth noise image (zero mean, unit std, + 0.1 white noise) and K one-hot
    ellipsoid label channels painted in order."""
    g = torch.Generator().manual_seed(seed)
    s = size
    low = torch.randn(1, 1, s // 8 + 2, s // 8 + 2, s // 8 + 2, generator=g)
    img = torch.nn.functional.interpolate(low, size=(s, s, s), mode="trilinear", align_corners=False)[0, 0]
    img = (img - img.mean()) / img.std() + 0.1 * torch.randn(s, s, s, generator=g)
    lab = torch.zeros(s, s, s, dtype=torch.int64)
    ax = torch.arange(s, dtype=torch.float32)
    for i in range(k):
        c = torch.rand(3, generator=g) * s
        r = 8 + 16 * torch.rand(3, generator=g) * (s / 160)
        m = (((ax[:, None, None] - c[0]) / r[0]) ** 2 + ((ax[None, :, None] - c[1]) / r[1]) ** 2 +
             ((ax[None, None, :] - c[2]) / r[2]) ** 2) <= 1.0
        lab[m] = i + 1
    onehot = torch.stack([(lab == i + 1).float() for i in range(k)])
    return torch.cat([img[None].float(), onehot]).contiguous()


def synthetic_label_mapping(k=15):
    """name_i -> (3*i, i): source ids spread over the 105 pretrain classes, target ids dense; background first."""
    names = ["background"] + [f"structure_{i:02d}" for i in range(1, k + 1)]
    return {n: (3 * i, i) for i, n in enumerate(names)}, names


def he_init_(model, seed=7):
    """nnU-Net's InitWeights_He(1e-2): kaiming_normal_(a=1e-2) for (transposed) conv weights, zero biases,
    InstanceNorm affine (1, 0).  Works on any module with .weight/.bias parameters laid out like PyTorch's."""
    g = torch.Generator().manual_seed(seed)
    gain = math.sqrt(2.0 / (1 + 1e-2 ** 2))
    with torch.no_grad():
        for m in model.modules():
            w = getattr(m, "weight", None)
            if w is None or w.dim() != 5:
                continue
            fan_in = w.shape[1] * w[0, 0].numel()
            w.copy_(torch.randn(w.shape, generator=g) * (gain / math.sqrt(fan_in)))
            m.bias.zero_()
    return model
