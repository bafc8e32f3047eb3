"""Raw-case preprocessing: the restated nnU-Net DefaultPreprocessor.  CPU: the oracle's pieces against
closed forms; GPU: the HIP resampling passes against scipy.ndimage.zoom (what skimage.resize, nnU-Net's resampler, calls)
and the whole run_case against the oracle, incl. the anisotropic separate-z branch and a NIfTI round trip."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
PLANS = json.loads((ROOT / "dg_tta_amd" / "__resources__" / "model_skeleton" / "plans.json").read_text())


def _case(seed=0, shape=(22, 26, 30)):
    rng = np.random.default_rng(seed)
    img = np.zeros((1, *shape), np.float32)
    img[:, 3:19, 4:22, 5:27] = rng.normal(-100, 400, (16, 18, 22))
    img[:, 8:12, 10:14, 12:16] = 0.0                       # a hole inside the body: filled by binary_fill_holes
    seg = np.zeros((1, *shape), np.int8)
    seg[:, 5:11, 6:13, 8:15] = 2
    seg[:, 10:15, 12:19, 14:21] = 5
    return img, seg


def test_oracle_pieces_cpu():
    from oracle import preprocessing as op
    img, seg = _case()
    d, s, bbox = op.crop_to_nonzero(img.copy(), seg.copy())
    assert bbox == [[3, 19], [4, 22], [5, 27]] and d.shape == (1, 16, 18, 22)
    assert (s == -1).sum() == 0                            # the hole is filled, nothing inside the box is "outside"
    assert op.compute_new_shape((16, 18, 22), (2.5, 0.9, 0.9), (1.5, 1.5, 1.5)) == [27, 11, 13]
    assert op.separate_z((5.0, 0.9, 0.9), (1.5, 1.5, 1.5)) == (True, 0)
    assert op.separate_z((2.5, 0.9, 0.9), (1.5, 1.5, 1.5)) == (False, None)
    # order-1 resize of a linear ramp is the ramp sampled at pixel centres (inside the clamped range)
    ramp = np.arange(10, dtype=float)
    out = op.resize(ramp, (20,), 1)
    x = (np.arange(20) + 0.5) * 0.5 - 0.5
    assert np.allclose(out, np.clip(x, 0, 9))
    # CT normalisation: clip to the plans' percentiles, then (x - mean) / std
    p = PLANS["foreground_intensity_properties_per_channel"]["0"]
    n = op.normalize(np.array([[[[-5000.0, 0.0, 5000.0]]]], np.float32), None, ["CTNormalization"], [False],
                     PLANS["foreground_intensity_properties_per_channel"])
    assert np.allclose(n.ravel(), [(p["percentile_00_5"] - p["mean"]) / p["std"], (0 - p["mean"]) / p["std"],
                                   (p["percentile_99_5"] - p["mean"]) / p["std"]], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("order", [0, 1, 3])
def test_resample_passes_match_scipy_zoom(order):
    from scipy import ndimage as ndi
    from dg_tta_amd import ops
    rng = np.random.default_rng(order)
    for shape, new in (((2, 13, 17, 21), (20, 9, 33)), ((1, 40, 6, 25), (40, 15, 25)), ((3, 5, 5, 5), (11, 2, 7))):
        vol = rng.normal(0, 100, shape)
        ref = np.stack([ndi.zoom(v, [n / o for n, o in zip(new, v.shape)], order=order, mode="nearest", grid_mode=True)
                        for v in vol])
        out = ops.resize_volume(torch.from_numpy(vol).to("cuda:0"), new, order).cpu().numpy()
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() < 1e-9 * np.abs(ref).max(), (shape, new, np.abs(out - ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("spacing", [(2.5, 0.9, 0.9), (5.0, 0.9, 0.9), (1.5, 1.5, 1.5)])
def test_run_case_matches_oracle(spacing):
    from dg_tta_amd.tta import preprocessing as pp
    from oracle import preprocessing as op
    img, seg = _case(1)
    d, s, props = pp.run_case_npy(img.copy(), seg.copy(), spacing, PLANS, "3d_fullres", "cuda:0")
    rd, rs, rprops = op.run_case_npy(img.copy(), seg.copy(), spacing, PLANS, "3d_fullres")
    assert d.shape == rd.shape and d.dtype == np.float32 and s.dtype == rs.dtype
    assert props["bbox_used_for_cropping"] == rprops["bbox_used_for_cropping"]
    assert np.abs(d - rd).max() < 2e-6 * max(1.0, np.abs(rd).max())
    assert (s == rs).mean() > 0.9999            # label borders sit on a 0.5 threshold of an interpolated indicator
    assert set(np.unique(s)) <= {-1, 0, 2, 5}


@pytest.mark.gpu
@pytest.mark.parametrize("ext", [".nii.gz", ".nrrd", ".mha"])
def test_nifti_case_through_load_tta_data(tmp_path, ext):
    """imagesTs/<case>_0000<ext> + labelsTs/<case><ext> -> {"data": [1+K, ...]} as preprocess_fromfile yields it; round 5: NRRD
    and MetaImage cases (tta/image_io.py) go through the same preprocessing as NIfTI ones."""
    from types import SimpleNamespace
    from dg_tta_amd.tta.image_io import write_image as write_nifti
    from dg_tta_amd.tta.nnunet_utils import load_tta_data
    from oracle import preprocessing as op
    img, seg = _case(2)
    (tmp_path / "imagesTs").mkdir()
    (tmp_path / "labelsTs").mkdir()
    write_nifti(tmp_path / "imagesTs" / f"case7_0000{ext}", img[0], spacing=(0.9, 0.9, 2.5))     # (x, y, z) spacing
    write_nifti(tmp_path / "labelsTs" / f"case7{ext}", seg[0].astype(np.int16), spacing=(0.9, 0.9, 2.5))
    predictor = SimpleNamespace(plans=PLANS, configuration="3d_fullres", device="cuda:0")
    cfg = {"tta_data_filepaths": [str(tmp_path / "imagesTs" / f"case7_0000{ext}")]}
    it, n = load_tta_data(cfg, tmp_path, predictor)
    items = list(it)
    assert n == 1 and items[0]["ofile"] == "tta_outputTs/case7"
    data = items[0]["data"]
    rd, rs, _ = op.run_case_npy(img.copy(), seg.astype(np.int16), (2.5, 0.9, 0.9), PLANS, "3d_fullres")
    assert tuple(data.shape) == (1 + 5, *rd.shape[1:])                      # image + one-hot channels for labels 1..5
    assert (data[0].numpy() - rd[0]).__abs__().max() < 2e-6 * np.abs(rd).max()
    lab = torch.cat([(data[1:].sum(0, keepdim=True) < 1).float(), data[1:]]).argmax(0).numpy()
    assert (lab == np.maximum(rs[0], 0)).mean() > 0.9999


@pytest.mark.gpu
@pytest.mark.parametrize("spacing", [(2.5, 0.9, 0.9), (5.0, 0.9, 0.9), (1.5, 1.5, 1.5)])
def test_export_in_original_geometry_matches_oracle(spacing):
    """The inverse of run_case for a prediction: accumulated window logits -> nnU-Net's
    convert_predicted_logits_to_segmentation_with_correct_shape (resample back, argmax, un-crop, transpose back) on the
    GPU class group by class group, against the scipy restatement, on a cropped case incl. the anisotropic branch."""
    from dg_tta_amd.tta import preprocessing as pp
    from dg_tta_amd.tta.inference import export_segmentation
    from oracle import inference as oinf
    img, seg = _case(3)
    d, s, props = pp.run_case_npy(img.copy(), seg.copy(), spacing, PLANS, "3d_fullres", "cuda:0")
    X, Y, Z = d.shape[1:]
    C = 19                                   # not a multiple of the export class group
    g = torch.Generator().manual_seed(11)
    # smooth-ish logits with clear winners: low-resolution noise upsampled, so that argmax regions have extent
    low = torch.randn(1, C, X // 3 + 2, Y // 3 + 2, Z // 3 + 2, generator=g)
    logits = torch.nn.functional.interpolate(low, size=(X, Y, Z), mode="trilinear", align_corners=False)[0] * 4
    nsum = torch.rand(X, Y, Z, generator=g) * 3 + 0.5            # per-voxel window weights (must be divided out first)
    acc = (logits * nsum).permute(1, 2, 3, 0).contiguous().to("cuda:0")
    crop = [slice(0, X), slice(0, Y), slice(0, Z)]
    out = export_segmentation(acc, nsum.to("cuda:0"), crop, props, PLANS, "3d_fullres")
    ref_logits = (acc.cpu() / nsum[..., None]).permute(3, 0, 1, 2).numpy()
    ref = oinf.convert_logits_to_segmentation_with_correct_shape(ref_logits, props, PLANS, "3d_fullres")
    assert out.shape == ref.shape == img.shape[1:] and out.dtype == ref.dtype
    bbox = props["bbox_used_for_cropping"]
    outside = np.ones(out.shape, bool)
    outside[tuple(slice(a, b) for a, b in bbox)] = False
    assert (out[outside] == 0).all()                               # un-cropped: zeros outside the box
    assert (out == ref).mean() > 0.9995                            # float ties at region borders only
    assert len(np.unique(out)) > 5


@pytest.mark.gpu
@pytest.mark.parametrize("spacing", [(5.0, 0.9, 0.9), (1.5, 1.5, 1.5)])
def test_export_from_feature_accumulators_matches_oracle(spacing):
    """Round 5: the same export from FEATURE-space window accumulators of a two-member ensemble (the head applied class group by
    class group in double, dgtta_feature_logits_chunk_f64) against the scipy restatement fed with the logits those features stand
    for; without resampling the fused head + argmax kernel gives the label map."""
    from dg_tta_amd.tta import preprocessing as pp
    from dg_tta_amd.tta.inference import WindowFeatures, export_segmentation
    from oracle import inference as oinf
    img, seg = _case(3)
    d, s, props = pp.run_case_npy(img.copy(), seg.copy(), spacing, PLANS, "3d_fullres", "cuda:0")
    X, Y, Z = d.shape[1:]
    M, C = 2, 19
    g = torch.Generator().manual_seed(12)
    low = torch.randn(M, 32, X // 3 + 2, Y // 3 + 2, Z // 3 + 2, generator=g)
    feat = torch.nn.functional.interpolate(low, size=(X, Y, Z), mode="trilinear", align_corners=False)      # [M,32,X,Y,Z]
    nsum = torch.rand(X, Y, Z, generator=g) * 3 + 0.5
    facc = (feat * nsum).permute(0, 2, 3, 4, 1).contiguous().to("cuda:0")
    w = torch.randn(M, C, 32, generator=g)
    b = torch.randn(M, C, generator=g)
    crop = [slice(0, X), slice(0, Y), slice(0, Z)]
    feats = WindowFeatures(facc, nsum.to("cuda:0"), crop, w.to("cuda:0"), b.to("cuda:0"))
    out = export_segmentation(feats, feats.nsum, crop, props, PLANS, "3d_fullres")
    ref_logits = (torch.einsum("mkxyz,mck->cxyz", feat.double(), w.double()) + b.double().sum(0)[:, None, None, None]).numpy()
    ref = oinf.convert_logits_to_segmentation_with_correct_shape(ref_logits, props, PLANS, "3d_fullres")
    assert out.shape == ref.shape == img.shape[1:] and out.dtype == ref.dtype
    assert (out == ref).mean() > 0.9995 and len(np.unique(out)) > 5
    plain = export_segmentation(feats, feats.nsum, crop, None, None, None)
    top2 = np.sort(ref_logits, 0)[-2:]
    safe = (top2[1] - top2[0]) > 1e-4 * np.abs(ref_logits).max()
    assert np.array_equal(plain[safe], ref_logits.argmax(0)[safe]) and safe.mean() > 0.99


@pytest.mark.gpu
def test_export_with_padded_small_case_and_no_resampling():
    """A case smaller than the patch (padded for the windows, cropped back) at the plans' spacing: export = argmax of the
    cropped accumulator pasted into the crop box."""
    from dg_tta_amd.tta.inference import export_segmentation
    X, Y, Z, C = 10, 16, 12, 7
    g = torch.Generator().manual_seed(2)
    acc = torch.randn(16, 16, 16, C, generator=g).to("cuda:0")
    nsum = torch.ones(16, 16, 16).to("cuda:0")
    crop = [slice(3, 13), slice(0, 16), slice(2, 14)]
    props = {"shape_before_cropping": (14, 18, 12), "bbox_used_for_cropping": [[2, 12], [1, 17], [0, 12]],
             "shape_after_cropping_and_before_resampling": (X, Y, Z), "spacing": [1.5, 1.5, 1.5]}
    out = export_segmentation(acc, nsum, crop, props, PLANS, "3d_fullres")
    ref = np.zeros((14, 18, 12), np.uint8)
    ref[2:12, 1:17, 0:12] = acc.cpu()[3:13, :, 2:14].argmax(-1).numpy()
    assert np.array_equal(out, ref.transpose(PLANS["transpose_backward"]))
    plain = export_segmentation(acc, nsum, crop, None, None, None)
    assert np.array_equal(plain, acc.cpu()[3:13, :, 2:14].argmax(-1).numpy())
