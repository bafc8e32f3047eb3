"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of nnU-Net's per-case segmentation metrics
[3P nnunetv2==2.2.1 evaluation/evaluate_predictions.py: compute_metrics / compute_tp_fp_fn_tn], used by the reference at
dg_tta/tta/tta.py:461-468.  Parity unpinned (nnunetv2 is not vendored with the reference): restated from the published
algorithm and checked on hand-computed cases in tests/test_formats.py."""
import numpy as np


def case_metrics(pred, ref, labels):
    out = {}
    for lab in labels:
        mp, mr = pred == lab, ref == lab
        tp = int(np.sum(mp & mr))
        fp = int(np.sum(mp & ~mr))
        fn = int(np.sum(~mp & mr))
        tn = int(np.sum(~mp & ~mr))
        if tp + fp + fn == 0:
            dice = iou = float("nan")
        else:
            dice, iou = 2 * tp / (2 * tp + fp + fn), tp / (tp + fp + fn)
        out[int(lab)] = {"Dice": dice, "IoU": iou, "FP": fp, "TP": tp, "FN": fn, "TN": tn, "n_pred": fp + tp, "n_ref": fn + tp}
    return out
