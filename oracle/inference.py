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


def convert_logits_to_segmentation_with_correct_shape(logits, properties, plans, configuration):
    """nnunetv2==2.2.1 inference/export_prediction.py: convert_predicted_logits_to_segmentation_with_correct_shape
    [third-party; the reference reaches it through predict_from_data_iterator,
    /root/reference/dg_tta/tta/nnunet_utils.py:208-230], restated from its published behaviour (parity unpinned):
    logits [C,X,Y,Z] (numpy / tensor, preprocessed geometry) -> resampling_fn_probabilities (resample_data_or_seg_to_shape:
    is_seg False, order 1, order_z 0, separate-z decided from the spacings) to shape_after_cropping_and_before_resampling
    -> softmax -> argmax -> zeros(shape_before_cropping)[bbox] = seg -> transpose(transpose_backward)."""
    from . import preprocessing as op
    conf = plans["configurations"][configuration]
    logits = np.asarray(logits, dtype=np.float32)
    cur_spacing = list(conf["spacing"])
    if len(cur_spacing) < len(properties["shape_after_cropping_and_before_resampling"]):
        cur_spacing = [properties["spacing"][0]] + cur_spacing
    kw = conf.get("resampling_fn_probabilities_kwargs", {"order": 1, "order_z": 0})
    do_sep, axis = op.separate_z(cur_spacing, properties["spacing"])
    res = op.resample_data_or_seg(logits, properties["shape_after_cropping_and_before_resampling"], False, axis,
                                  kw["order"], do_sep, kw["order_z"])
    probs = torch.softmax(torch.from_numpy(np.ascontiguousarray(res)).float(), 0)
    seg = probs.argmax(0).numpy()
    full = np.zeros(properties["shape_before_cropping"], dtype=np.uint8 if logits.shape[0] - 1 < 255 else np.uint16)
    sl = tuple(slice(int(a), int(b)) for a, b in properties["bbox_used_for_cropping"])
    full[sl] = seg
    return full.transpose(plans["transpose_backward"])
