"""Folder evaluation of predicted label maps — the step after ensemble inference (dg_tta/tta/tta.py:447-470), which the
reference delegates to nnU-Net's compute_metrics_on_folder_simple [3P nnunetv2==2.2.1, evaluation/evaluate_predictions.py].
Restated from the published algorithm (parity unpinned: nnunetv2 is not vendored with the reference): per case and label
TP / FP / FN / TN, Dice = 2TP/(2TP+FP+FN), IoU = TP/(TP+FP+FN) (NaN when the label is absent from both), n_pred, n_ref;
'mean' = nanmean over cases, 'foreground_mean' = mean over labels != 0.  The per-label counts come from the HIP
label-count kernel (dgtta_argmax_dice), the arithmetic on them is host-side."""
import json
from pathlib import Path

import numpy as np
import torch

from .. import ops
from .image_io import is_image_file, read_image


def load_label_map(path):
    p = str(path)
    if p.endswith(".npy"):
        return np.load(p)
    if is_image_file(p):
        return read_image(p)[0]
    raise ValueError(f"unsupported label map format: {p}")


def case_metrics(pred, ref, labels, device="cuda"):
    """metrics[label] = {Dice, IoU, FP, TP, FN, TN, n_pred, n_ref} for two integer label maps of equal shape."""
    if tuple(pred.shape) != tuple(ref.shape):
        raise ValueError(f"shape mismatch: prediction {tuple(pred.shape)} vs reference {tuple(ref.shape)}")
    nlab = int(max(labels)) + 1
    p = torch.as_tensor(np.ascontiguousarray(pred).astype(np.int64)).to(device)
    r = torch.as_tensor(np.ascontiguousarray(ref).astype(np.int64)).to(device)
    _, counts = ops.argmax_dice_from_labels(p, r, nlab)        # rows: n_pred, n_ref, TP
    c = counts.cpu().numpy().astype(np.int64)
    total = int(p.numel())
    out = {}
    for lab in labels:
        n_pred, n_ref, tp = int(c[0, lab]), int(c[1, lab]), int(c[2, lab])
        fp, fn = n_pred - tp, n_ref - tp
        tn = total - tp - fp - fn
        if tp + fp + fn == 0:
            dice = iou = float("nan")
        else:
            dice, iou = 2 * tp / (2 * tp + fp + fn), tp / (tp + fp + fn)
        out[int(lab)] = {"Dice": dice, "IoU": iou, "FP": fp, "TP": tp, "FN": fn, "TN": tn, "n_pred": n_pred, "n_ref": n_ref}
    return out


def compute_metrics_on_folder_simple(folder_ref, folder_pred, labels, output_file=None, device="cuda",
                                     suffixes=(".nii.gz", ".nii", ".nrrd", ".mha", ".mhd", ".npy")):
    """Evaluates every prediction in folder_pred that has a reference of the same name in folder_ref; returns the summary
    dict (and writes it as JSON to output_file) in nnU-Net's layout: metric_per_case / mean / foreground_mean."""
    folder_ref, folder_pred = Path(folder_ref), Path(folder_pred)
    files = sorted(f for f in folder_pred.iterdir() if f.name.endswith(tuple(suffixes)))
    per_case = []
    for f in files:
        ref = folder_ref / f.name
        if not ref.is_file():
            continue
        m = case_metrics(load_label_map(f), load_label_map(ref), labels, device)
        per_case.append({"metrics": m, "prediction_file": str(f), "reference_file": str(ref)})
    keys = ["Dice", "IoU", "FP", "TP", "FN", "TN", "n_pred", "n_ref"]
    means = {}
    for lab in labels:
        means[int(lab)] = {}
        for k in keys:
            vals = np.array([c["metrics"][int(lab)][k] for c in per_case], dtype=np.float64)
            means[int(lab)][k] = float(np.nanmean(vals)) if vals.size and not np.all(np.isnan(vals)) else float("nan")
    fg = [l for l in labels if l != 0]
    foreground_mean = {k: float(np.mean([means[int(l)][k] for l in fg])) if fg else float("nan") for k in keys}
    summary = {"metric_per_case": per_case, "mean": means, "foreground_mean": foreground_mean}
    if output_file is not None:
        Path(output_file).parent.mkdir(parents=True, exist_ok=True)
        with open(output_file, "w") as fh:
            json.dump(_jsonable(summary), fh, indent=4, sort_keys=False)
    return summary


def _jsonable(o):
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    return o
