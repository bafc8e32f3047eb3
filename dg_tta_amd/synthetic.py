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


# ------------------------------------------------------------------------------------------------ a learnable task
# Round 5: synthetic_case() paints labels that are INDEPENDENT of the image, so no network can learn them
# and every Dice figure measured on it is ~0.  The "atlas" cases below are a task a network CAN learn, with a source
# domain to pre-train on and a shifted target domain to adapt to - what DG-TTA is for (CT-trained model, MR target).
def _lowfreq(size, cells, gen):
    """Smooth random field in [-1, 1]-ish: trilinear upsampling of white noise on a `cells`^3 lattice."""
    low = torch.randn(1, 1, cells + 2, cells + 2, cells + 2, generator=gen)
    return torch.nn.functional.interpolate(low, size=(size, size, size), mode="trilinear", align_corners=False)[0, 0]


def atlas_layout(k, atlas_seed=11):
    """K ellipsoids of a fixed 'anatomy' in unit coordinates: centres in [0.2, 0.8]^3 kept >= 0.22 apart where that is
    possible, radii in [0.09, 0.17] per axis, one mean intensity per structure (distinct levels, shuffled)."""
    g = torch.Generator().manual_seed(atlas_seed)
    centres = []
    for i in range(k):
        best, best_d = None, -1.0
        for _ in range(64):            # best of 64 candidates: the farthest from the structures placed so far
            c = 0.2 + 0.6 * torch.rand(3, generator=g)
            d = min([float((c - o).norm()) for o in centres], default=1.0)
            if d > best_d:
                best, best_d = c, d
            if d >= 0.22:
                break
        centres.append(best)
    radii = 0.09 + 0.08 * torch.rand(k, 3, generator=g)
    levels = torch.linspace(-1.6, 1.6, k)[torch.randperm(k, generator=g)]
    return torch.stack(centres), radii, levels


def atlas_case(size, k, seed, domain="source", atlas_seed=11, noise=None, thick=2):
    """One case of the atlas task: data [1+K, S, S, S] fp32 (image + K one-hot label channels, the layout get_batch reads).

    Anatomy: the K ellipsoids of atlas_layout() with per-case jitter (centres +- 0.03, radii x U(0.85, 1.15)), painted in
    order.  Source appearance ('CT-like'): each structure at its own intensity level over a smooth background, mild
    texture, 3-tap smoothing, white noise 0.05.  Target appearance (domain='target', 'MR-like'): the SAME anatomy model
    under inverted, gamma-remapped contrast, a multiplicative low-frequency bias field (0.6 .. 1.4), thick slices along the
    first axis (`thick`-voxel averaging, as a 3 mm acquisition of 1.5 mm anatomy) and white noise `noise` (default 0.12; 0.5
    is the 'low-SNR' target on which a MIND-only model loses a fifth of its Dice and test-time adaptation wins part of it
    back, tests/golden/make_golden_r5.py).  `noise` also overrides the source domain's 0.05."""
    assert domain in ("source", "target")
    g = torch.Generator().manual_seed(1000003 * (1 if domain == "source" else 2) + seed)
    centres, radii, levels = atlas_layout(k, atlas_seed)
    s = size
    ax = (torch.arange(s, dtype=torch.float32) + 0.5) / s
    lab = torch.zeros(s, s, s, dtype=torch.int64)
    for i in range(k):
        c = centres[i] + 0.06 * (torch.rand(3, generator=g) - 0.5)
        r = radii[i] * (0.85 + 0.3 * torch.rand(3, generator=g))
        m = (((ax[:, None, None] - c[0]) / r[0]) ** 2 + ((ax[None, :, None] - c[1]) / r[1]) ** 2 +
             ((ax[None, None, :] - c[2]) / r[2]) ** 2) <= 1.0
        lab[m] = i + 1
    cells = max(2, s // 8)
    img = 0.35 * _lowfreq(s, cells, g)
    tex = _lowfreq(s, max(3, s // 4), g)
    for i in range(k):
        m = lab == i + 1
        img[m] = levels[i] + 0.15 * tex[m]
    img = torch.nn.functional.avg_pool3d(img[None, None], 3, 1, 1, count_include_pad=False)[0, 0]
    if domain == "source":
        img = img + (0.05 if noise is None else noise) * torch.randn(s, s, s, generator=g)
    else:
        img = -img
        img = torch.sign(img) * img.abs().pow(0.7)
        img = img * (1.0 + 0.4 * _lowfreq(s, 2, g).clamp(-1, 1))
        if thick > 1:
            slab = torch.nn.functional.avg_pool3d(img[None, None], (thick, 1, 1), (thick, 1, 1), ceil_mode=True)
            img = slab.repeat_interleave(thick, dim=2)[0, 0, :s]
        img = img + (0.12 if noise is None else noise) * torch.randn(s, s, s, generator=g)
    onehot = torch.stack([(lab == i + 1).float() for i in range(k)])
    return torch.cat([img[None].float(), onehot]).contiguous()
