"""Sliding-window ensemble inference, CPU restatement (oracle; test infrastructure).

Restates nnunetv2==2.2.1's `predict_sliding_window_return_logits` / `compute_gaussian` /
`compute_steps_for_sliding_window` [third-party, not under /root/reference; reached from
dg_tta/tta/nnunet_utils.py:116-125,208-230 and dg_tta/tta/tta.py:379-416] from their published behaviour.
parity unpinned: there is neither a vendored copy nor a reference test for this stage.
"""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def compute_gaussian(tile_size, sigma_scale=1.0 / 8, value_scaling_factor=10.0):
    tmp = np.zeros(tile_size)
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = torch.from_numpy(gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode="constant", cval=0)).float()
    g = g / g.max() * value_scaling_factor
    g[g == 0] = g[g != 0].min()
    return g


def steps_1d(image, tile, step=0.5):
    n = int(np.ceil((image - tile) / (tile * step))) + 1
    if n == 1:
        return [0]
    actual = (image - tile) / (n - 1)
    return [int(np.round(actual * i)) for i in range(n)]


@torch.no_grad()
def ensemble_logits(models, data, patch_size):
    """models: list of callables [1,C,P] -> [1,ncls,P]; data [C,X,Y,Z] with every dim >= patch."""
    X, Y, Z = data.shape[1:]
    g = compute_gaussian(tuple(patch_size))
    total = None
    for m in models:
        acc, n = None, torch.zeros(X, Y, Z)
        for sx in steps_1d(X, patch_size[0]):
            for sy in steps_1d(Y, patch_size[1]):
                for sz in steps_1d(Z, patch_size[2]):
                    sl = (slice(sx, sx + patch_size[0]), slice(sy, sy + patch_size[1]), slice(sz, sz + patch_size[2]))
                    out = m(data[(slice(None),) + sl][None])[0]
                    if acc is None:
                        acc = torch.zeros(out.shape[0], X, Y, Z)
                    acc[(slice(None),) + sl] += out * g
                    n[sl] += g
        logits = acc / n
        total = logits if total is None else total + logits
    return total / len(models)
