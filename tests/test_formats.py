"""On-disk formats next to the hot path: NIfTI-1 label-map interchange and the folder evaluation."""
import gzip
import json
import struct

import numpy as np
import pytest
import torch

from dg_tta_amd.tta.nifti_io import read_nifti, write_nifti


def test_nifti_round_trip_gz_and_geometry(tmp_path):
    rng = np.random.default_rng(3)
    seg = rng.integers(0, 9, size=(5, 7, 11)).astype(np.uint8)           # [z,y,x]
    p = tmp_path / "case_000.nii.gz"
    write_nifti(p, seg, spacing=(1.5, 0.75, 3.0))
    back, hdr = read_nifti(p)
    assert back.dtype == np.uint8 and np.array_equal(back, seg)
    assert hdr["pixdim"] == (1.5, 0.75, 3.0) and hdr["shape_xyz"] == [11, 7, 5]
    assert np.allclose(np.diag(hdr["affine"])[:3], (1.5, 0.75, 3.0))
    raw = gzip.open(p, "rb").read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\0" and len(raw) == 352 + seg.size
    # x is the fastest axis on disk (NIfTI), z the slowest
    assert raw[352 + 1] == seg[0, 0, 1] and raw[352 + 11] == seg[0, 1, 0]
    # re-write a prediction with the reference geometry: the header travels
    q = tmp_path / "pred.nii"
    write_nifti(q, (seg.astype(np.int16) + 1), header=hdr)
    b2, h2 = read_nifti(q)
    assert b2.dtype == np.int16 and np.array_equal(b2, seg.astype(np.int16) + 1) and h2["pixdim"] == hdr["pixdim"]
    with pytest.raises(ValueError):
        (tmp_path / "bad.nii").write_bytes(b"\0" * 400)
        read_nifti(tmp_path / "bad.nii")


def test_metric_oracle_on_a_hand_case():
    from oracle.evaluation import case_metrics
    pred = np.array([[0, 1, 1, 2], [0, 1, 2, 2]])
    ref = np.array([[0, 1, 2, 2], [1, 1, 2, 0]])
    m = case_metrics(pred, ref, [0, 1, 2, 3])
    assert (m[1]["TP"], m[1]["FP"], m[1]["FN"], m[1]["TN"]) == (2, 1, 1, 4)
    assert m[1]["Dice"] == pytest.approx(4 / 6) and m[1]["IoU"] == pytest.approx(0.5)
    assert (m[2]["TP"], m[2]["FP"], m[2]["FN"]) == (2, 1, 1) and m[2]["n_pred"] == 3 and m[2]["n_ref"] == 3
    assert np.isnan(m[3]["Dice"]) and m[3]["TN"] == 8


@pytest.mark.gpu
def test_folder_evaluation_matches_oracle(tmp_path):
    from dg_tta_amd.tta.evaluation import compute_metrics_on_folder_simple
    from oracle.evaluation import case_metrics
    rng = np.random.default_rng(11)
    labels = list(range(6))
    (tmp_path / "ref").mkdir()
    (tmp_path / "pred").mkdir()
    cases = {}
    for i, ext in enumerate([".nii.gz", ".npy", ".nii.gz"]):
        ref = rng.integers(0, 5, size=(9, 14, 17)).astype(np.int16)      # label 5 absent everywhere -> NaN Dice
        pred = np.where(rng.random(ref.shape) < 0.8, ref, rng.integers(0, 5, size=ref.shape)).astype(np.int16)
        name = f"case_{i:03d}{ext}"
        for folder, arr in (("ref", ref), ("pred", pred)):
            if ext == ".npy":
                np.save(tmp_path / folder / name, arr)
            else:
                write_nifti(tmp_path / folder / name, arr, spacing=(1.5, 1.5, 1.5))
        cases[name] = (pred, ref)
    np.save(tmp_path / "pred" / "no_reference.npy", np.zeros((2, 2, 2), np.int16))
    out = tmp_path / "summary_Ts.json"
    summary = compute_metrics_on_folder_simple(tmp_path / "ref", tmp_path / "pred", labels, output_file=out)
    assert len(summary["metric_per_case"]) == 3
    for c in summary["metric_per_case"]:
        name = c["prediction_file"].split("/")[-1]
        want = case_metrics(*cases[name], labels)
        for lab in labels:
            for k, v in want[lab].items():
                got = c["metrics"][lab][k]
                assert (np.isnan(v) and np.isnan(got)) or got == pytest.approx(v), (name, lab, k)
    for lab in labels[:5]:
        dices = [case_metrics(*cases[n], labels)[lab]["Dice"] for n in cases]
        assert summary["mean"][lab]["Dice"] == pytest.approx(np.nanmean(dices))
    on_disk = json.loads(out.read_text())
    assert set(on_disk) == {"metric_per_case", "mean", "foreground_mean"} and "1" in on_disk["mean"]
    assert on_disk["foreground_mean"]["TP"] == pytest.approx(summary["foreground_mean"]["TP"])
