"""MIND3D self-similarity descriptor, CPU restatement (oracle; test infrastructure).

Follows /root/reference/dg_tta/mind.py:97-168 (MIND3D), :27-43 (smooth), :5-24 (filter1D).
The two one-hot 3x3x3 "shift" convolutions of the reference (mind.py:145-148) select one
voxel of the replicate-padded image each, so they are restated as gathers; adding the
26 zero products of the reference's conv is exact in fp32, so the result is bit-equal.
"""
import torch
import torch.nn.functional as F

# (d,h,w) positions inside the 3x3x3 window, dumped from mshift1 / mshift2
# (mind.py:112-135: pairs of the six-neighbourhood with squared distance 2, x>y).
SHIFT1 = [(1, 1, 0), (1, 0, 1), (1, 0, 1), (1, 1, 2), (1, 1, 2), (2, 1, 1),
          (2, 1, 1), (2, 1, 1), (1, 2, 1), (1, 2, 1), (1, 2, 1), (1, 2, 1)]
SHIFT2 = [(0, 1, 1), (0, 1, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0),
          (1, 0, 1), (1, 1, 2), (0, 1, 1), (1, 1, 0), (1, 1, 2), (2, 1, 1)]


def gauss_taps(sigma=1.0):
    """mind.py:30-37: N = 2*ceil(1.5 sigma)+1 taps, exp(-x^2/2s^2), normalised."""
    s = torch.tensor([float(sigma)])
    n = int(torch.ceil(s * 3.0 / 2.0).long().item()) * 2 + 1
    w = torch.exp(-torch.pow(torch.linspace(-(n // 2), n // 2, n), 2) / (2 * torch.pow(s, 2)))
    return w / w.sum()


def _filter_axis(vol, taps, axis):
    """mind.py:5-24: replicate-padded 1-D correlation along spatial axis 0/1/2, per channel."""
    b, c, d, h, w = vol.shape
    n = taps.numel()
    pad = [0] * 6
    pad[4 - 2 * axis] = pad[5 - 2 * axis] = n // 2
    shape = [1, 1, 1, 1, 1]
    shape[axis + 2] = n
    x = F.pad(vol.reshape(b * c, 1, d, h, w), pad, mode="replicate")
    return F.conv3d(x, taps.view(shape)).view(b, c, d, h, w)


def smooth(vol, sigma=1.0):
    taps = gauss_taps(sigma)
    for axis in (0, 1, 2):
        vol = _filter_axis(vol, taps, axis)
    return vol


def edge_selection(img, delta=1):
    """mind.py:145-148 as gathers on the replicate-padded image -> [B,12,D,H,W]."""
    b, _, d, h, w = img.shape
    p = F.pad(img, [delta] * 6, mode="replicate")

    def pick(pos):
        z, y, x = (q * delta for q in pos)
        return p[:, 0, z:z + d, y:y + h, x:x + w]

    return torch.stack([pick(a) - pick(c) for a, c in zip(SHIFT1, SHIFT2)], dim=1)


def mind3d(img, noise, delta=1, sigma=1.0, randn_weighting=0.05):
    """MIND3D.forward (mind.py:142-164) with the randn_like draw passed in as `noise`."""
    e = edge_selection(img, delta)
    e = e + randn_weighting * noise
    ssd = smooth(e ** 2, sigma)
    mind = ssd - torch.min(ssd, 1, keepdim=True)[0]
    var = torch.mean(mind, 1, keepdim=True)
    gm = var.mean()
    var = torch.clamp(var, gm * 0.001, gm * 1000)
    mind = mind / var
    return torch.exp(-mind)


def mind3d_seeded(img):
    """Same draw as the reference's `torch.randn_like(edge_selection)` on the CPU generator."""
    b, _, d, h, w = img.shape
    return mind3d(img, torch.randn(b, 12, d, h, w))
