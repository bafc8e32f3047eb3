"""Sliding-window ensemble inference ("next" row of SURVEY.md §8f): window arithmetic on the CPU, parity of the HIP
path against the CPU restatement on the GPU."""
import numpy as np
import pytest
import torch

DEV = "cuda:0"


def test_window_steps_and_gaussian_cpu():
    from dg_tta_amd.tta.inference import compute_gaussian, compute_steps_for_sliding_window, pad_to_patch
    from oracle import inference as oinf
    assert compute_steps_for_sliding_window((512, 512, 512), (128, 128, 128)) == [[0, 64, 128, 192, 256, 320, 384]] * 3
    assert compute_steps_for_sliding_window((128, 160, 130), (128, 128, 128)) == [[0], [0, 32], [0, 2]]
    assert compute_steps_for_sliding_window((231, 228, 242), (112, 112, 128)) == [[0, 40, 79, 119], [0, 39, 77, 116], [0, 57, 114]]
    for image, tile in ((40, 16), (16, 16), (231, 112)):
        assert oinf.steps_1d(image, tile) == compute_steps_for_sliding_window((image,) * 3, (tile,) * 3)[0]
    g = compute_gaussian((16, 16, 16))
    assert torch.equal(g, oinf.compute_gaussian((16, 16, 16)))
    assert float(g.max()) == 10.0 and float(g.min()) > 0 and g.argmax() == np.ravel_multi_index((8, 8, 8), (16, 16, 16))
    assert torch.allclose(g, g.flip(0).roll(1, 0), atol=1e-6)            # symmetric about the centre voxel
    x = torch.ones(1, 10, 20, 13)
    padded, crop = pad_to_patch(x, [16, 16, 16])
    assert tuple(padded.shape) == (1, 16, 20, 16) and padded[(slice(None),) + tuple(crop)].shape == x.shape
    assert float(padded.sum()) == float(x.sum())


def test_window_accumulator_storage_switch(monkeypatch):
    """DGTTA_WINDOW_ACC: fp32 (default) | fp16; anything else is refused before memory is allocated."""
    from dg_tta_amd.tta.inference import window_acc_dtype, _acc_code
    from dg_tta_amd import ops
    monkeypatch.delenv("DGTTA_WINDOW_ACC", raising=False)
    assert window_acc_dtype() is torch.float32
    monkeypatch.setenv("DGTTA_WINDOW_ACC", "FP16")
    assert window_acc_dtype() is torch.float16
    monkeypatch.setenv("DGTTA_WINDOW_ACC", "bf16")
    with pytest.raises(ValueError, match="fp32 or fp16"):
        window_acc_dtype()
    assert _acc_code(torch.empty(1, dtype=torch.float16)) == ops.F16 and _acc_code(torch.empty(1)) == ops.F32
    with pytest.raises(ValueError):
        _acc_code(torch.empty(1, dtype=torch.bfloat16))
    with pytest.raises(Exception):          # no CPU fallback: the label-map kernel refuses host tensors
        ops.argmax_rows(torch.zeros(4, 5))
    # the fused head + accumulate path validates the accumulator's storage type as the plain path does
    from dg_tta_amd.tta.inference import predict_sliding_window_return_logits
    from dg_tta_amd.unet import HipPlainConvUNet
    from conftest import SMALL_CFG
    net = HipPlainConvUNet(SMALL_CFG)
    for bad in (torch.bfloat16, torch.float64):
        with pytest.raises(ValueError, match="fp32 or fp16"):
            predict_sliding_window_return_logits(net, torch.zeros(1, 16, 16, 16), [16, 16, 16], acc_dtype=bad)
        with pytest.raises(ValueError, match="fp32 or fp16"):
            net.fuse_window_accumulate(torch.zeros(2, 2, 2, 9, dtype=bad), torch.zeros(2, 2, 2), torch.zeros(2, 2, 2), [])
    with pytest.raises(ValueError, match="acc_dtype"):
        predict_sliding_window_return_logits(net, torch.zeros(1, 16, 16, 16), [16, 16, 16], acc=torch.zeros(16, 16, 16, 9),
                                             acc_dtype=torch.float16)


# a net whose head reads 32 channels: the product then accumulates FEATURES and runs the head once per voxel (round 5)
FEAT32_CFG = dict(features=(32, 40), strides=(1, 2), n_conv_enc=(2, 2), n_conv_dec=(2,), in_channels=12, num_classes=9)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["small", "feat32"])
def test_ensemble_sliding_window_matches_cpu_restatement(tmp_path, cfg_name):
    from conftest import SMALL_CFG
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.tta import inference as pinf
    from dg_tta_amd.tta.inference import run_inference
    from dg_tta_amd.unet import HipPlainConvUNet
    from oracle import inference as oinf, mind as omind, unet as ounet
    SMALL_CFG = SMALL_CFG if cfg_name == "small" else FEAT32_CFG
    torch.manual_seed(0)
    data = torch.randn(1, 40, 36, 44)
    patch = [16, 16, 16]
    members = [ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(SMALL_CFG), s), s + 1) for s in (1, 2)]
    # MIND noise is drawn per window on the device generator in the product path; use zero noise weighting on both
    # sides to make the comparison deterministic (randn_weighting = 0 -> the draw does not matter).
    zeros = torch.zeros(1, 12, *patch)
    cpu_models = [lambda x, m=m: m(omind.mind3d(x, zeros, randn_weighting=0.0)) for m in members]
    ref = oinf.ensemble_logits(cpu_models, data, patch)
    net = HipPlainConvUNet(SMALL_CFG, conv_impl=1).to(DEV)
    net.register_forward_pre_hook(lambda mod, inp: MIND3D(randn_weighting=0.0).forward(*inp))
    with torch.no_grad():
        assert pinf._can_accumulate_features(net) == (cfg_name == "feat32")
    seg = run_inference(data, net, [m.state_dict() for m in members], patch)
    assert tuple(seg.shape) == (40, 36, 44) and seg.dtype == torch.int64
    if cfg_name == "feat32":               # the accumulated ensemble logits themselves, re-formed from the features
        feats = pinf.predict_ensemble_features(data, net, [m.state_dict() for m in members], patch)
        got = (feats.logits() / feats.nsum[..., None] / len(members))[tuple(feats.crop)].permute(3, 0, 1, 2).cpu()
        assert float((got - ref).abs().max()) < 2e-4 * float(ref.abs().max())
    ref_seg = ref.argmax(0)
    top2 = ref.topk(2, dim=0).values
    safe = (top2[0] - top2[1]) > 1e-3
    assert torch.equal(seg[safe], ref_seg[safe])                     # bit-exact labels outside float-tie voxels
    assert (seg == ref_seg).float().mean() > 0.999
    # mapping to target label ids (tta.py:407-411)
    mapping = {"background": (0, 0), "a": (2, 1), "b": (5, 2)}
    mapped = run_inference(data, net, [m.state_dict() for m in members], patch, mapping, ["background", "a", "b"])
    assert set(mapped.unique().tolist()) <= {0, 1, 2}
    assert torch.equal(mapped == 1, seg == 2) and torch.equal(mapped == 2, seg == 5)


@pytest.mark.gpu
def test_sliding_window_at_size_properties():
    """BASELINE config 3's mechanics at size: the FULL 3d_fullres net (105 classes, bf16 MFMA path) over a 256 x 192 x 320
    volume with 128^3 Gaussian windows (3 x 2 x 4 = 24 windows, batched 4 per pass).  Size-independent properties:
    (a) the accumulated weight map equals the sum of the window Gaussians (computed on the host from the window origins);
    (b) acc / nsum inside a region covered by exactly ONE window equals the plain forward of that window;
    (c) a volume of exactly one window reproduces the plain forward everywhere (weights cancel);
    (d) the label map is the argmax of the accumulated logits over ALL pretrain classes."""
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.tta.inference import (compute_gaussian, compute_steps_for_sliding_window,
                                          predict_sliding_window_return_logits, run_inference)
    from dg_tta_amd.unet import HipPlainConvUNet
    net = he_init_(HipPlainConvUNet(act_dtype=torch.bfloat16), seed=7).to(DEV)
    net.register_forward_pre_hook(lambda mod, inp: MIND3D(randn_weighting=0.0).forward(*inp, out_dtype=torch.bfloat16))
    patch = [128, 128, 128]
    torch.manual_seed(1)
    vol = torch.randn(1, 256, 192, 320)
    acc, nsum, crop = predict_sliding_window_return_logits(net, vol, patch)
    assert tuple(acc.shape) == (256, 192, 320, 105) and all(c == slice(0, s) for c, s in zip(crop, vol.shape[1:]))
    steps = compute_steps_for_sliding_window(vol.shape[1:], patch)
    assert steps == [[0, 64, 128], [0, 64], [0, 64, 128, 192]]
    g = compute_gaussian(tuple(patch))
    ref_n = torch.zeros(vol.shape[1:])
    for sx in steps[0]:
        for sy in steps[1]:
            for sz in steps[2]:
                ref_n[sx:sx + 128, sy:sy + 128, sz:sz + 128] += g
    assert (nsum.cpu() - ref_n).abs().max() < 1e-4 * ref_n.max()                        # (a)
    # (b) the corner [0:64, 0:64, 0:64] is covered by the first window only
    with torch.no_grad():
        first = net(vol[None, :, :128, :128, :128].to(DEV)).float()[0]                  # [105,128,128,128]
    corner = (acc[:64, :64, :64] / nsum[:64, :64, :64, None]).permute(3, 0, 1, 2)
    # (batch-1 vs batch-4 passes pick different tilings for the small layers: bf16 rounding of the activations differs)
    assert (corner - first[:, :64, :64, :64]).abs().max() < 1.5e-2 * first.abs().max()
    # (d) label map = argmax over all 105 classes of the accumulated logits
    seg = run_inference(vol, net, [net.state_dict()], patch)
    am = acc.argmax(-1).cpu()
    top2 = acc.topk(2, dim=-1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-3 * nsum).cpu()
    assert torch.equal(seg[safe], am[safe]) and (seg == am).float().mean() > 0.999
    del acc, nsum, corner, top2
    # (c) single-window volume
    one = vol[:, :128, :128, :128]
    acc1, nsum1, _ = predict_sliding_window_return_logits(net, one, patch)
    assert (acc1 / nsum1[..., None] - first.permute(1, 2, 3, 0)).abs().max() < 1e-4 * first.abs().max()


@pytest.mark.gpu
def test_sliding_window_512_full_size_properties():
    """BASELINE config 3 at FULL size (round 4): 512^3 volume, 128^3 windows at step 0.5 = 7^3 = 343 windows, 105 classes,
    52.5 GiB fp32 accumulator.  Size-independent properties: the weight map equals the sum of the 343 window Gaussians
    (separable: the product of three 1-D sums, formed on the host), every voxel is covered, the label map is the argmax of
    the accumulator (streaming kernel against torch on a slab), and the fp16 accumulator gives the same labels wherever the
    fp32 top-2 margin exceeds its rounding bound."""
    from dg_tta_amd import ops
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.tta.inference import compute_gaussian, compute_steps_for_sliding_window, predict_sliding_window_return_logits
    from dg_tta_amd.unet import HipPlainConvUNet
    net = he_init_(HipPlainConvUNet(act_dtype=torch.bfloat16), seed=7)
    net.decoder.seg_layers[-1].bias.data.normal_()
    net.register_forward_pre_hook(lambda mod, inp: MIND3D(randn_weighting=0.0).forward(*inp, out_dtype=torch.bfloat16))
    net = net.to(DEV)
    n, patch = 512, [128, 128, 128]
    vol = torch.randn(1, n, n, n, generator=torch.Generator().manual_seed(5))
    steps = compute_steps_for_sliding_window((n, n, n), patch)
    assert steps == [[0, 64, 128, 192, 256, 320, 384]] * 3
    acc, nsum, crop = predict_sliding_window_return_logits(net, vol, patch, acc_dtype=torch.float32)
    assert tuple(acc.shape) == (n, n, n, 105) and acc.dtype == torch.float32
    # the Gaussian is NOT separable after nnU-Net's rescaling / zero replacement, so the reference map is summed window by
    # window - on the device, with torch (343 slice additions)
    g = compute_gaussian(tuple(patch)).to(DEV)
    ref_n = torch.zeros((n, n, n), device=DEV)
    for sx in steps[0]:
        for sy in steps[1]:
            for sz in steps[2]:
                ref_n[sx:sx + 128, sy:sy + 128, sz:sz + 128] += g
    assert float(nsum.min()) > 0
    assert float((nsum - ref_n).abs().max()) <= 1e-5 * float(ref_n.max())      # 8 fp32 additions per voxel, order aside
    del ref_n
    seg = ops.argmax_rows(acc)
    sl = slice(200, 232)
    assert torch.equal(seg[sl], acc[sl].argmax(-1))
    assert len(seg[sl].unique()) > 10
    top2 = acc[sl].topk(2, dim=-1).values
    lmax = float((acc[sl].abs().amax(-1) / nsum[sl]).max())
    # (a window's own logits may exceed the blended ones that lmax is taken from: factor 4 on the rounding bound of 8 windows)
    safe = (top2[..., 0] - top2[..., 1]) > 2 * 4 * 8 * 2.0 ** -11 * lmax * nsum[sl]
    safe_f = (top2[..., 0] - top2[..., 1]) > 1e-4 * lmax * nsum[sl]        # fp32 sums of 8 + 32 terms re-associated: ~1e-6
    seg32 = seg[sl].clone()
    del acc, seg, top2
    torch.cuda.empty_cache()
    acc16, nsum16, _ = predict_sliding_window_return_logits(net, vol, patch, acc_dtype=torch.float16)
    assert acc16.dtype == torch.float16 and torch.equal(nsum16, nsum)
    seg16 = ops.argmax_rows(acc16)[sl]
    assert torch.equal(seg16[safe], seg32[safe]) and float((seg16 == seg32).float().mean()) > 0.99 and float(safe.float().mean()) > 0.3
    del acc16, seg16
    torch.cuda.empty_cache()
    # round 5: the feature-space accumulator (16 GiB instead of 52.5): same weight map, same labels outside float ties
    from dg_tta_amd.tta.inference import WindowFeatures, accumulate_window_features
    facc, nsum_f, crop_f = accumulate_window_features(net, vol, patch)
    assert tuple(facc.shape) == (n, n, n, 32) and torch.equal(nsum_f, nsum)
    head = net.decoder.seg_layers[-1]
    feats = WindowFeatures(facc[None], nsum_f, crop_f, head.weight.detach().reshape(1, 105, 32).float(), head.bias.detach().float()[None])
    seg_f = feats.argmax()[sl]
    assert torch.equal(seg_f[safe_f], seg32[safe_f]) and float(safe_f.float().mean()) > 0.9
    assert float((seg_f == seg32).float().mean()) > 0.9999


@pytest.mark.gpu
@pytest.mark.parametrize("acc_dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C,rows", [(105, 64 * 40 + 17), (105, 4096), (16, 5000), (2, 4100), (112, 129), (7, 63),
                                    (118, 4097), (113, 50), (300, 70)])        # > 112 classes: the one-wave-per-row kernel
def test_argmax_rows_first_maximum_wins(C, rows, acc_dtype):
    """Round 4: the streaming label-map kernel (dgtta_argmax_rows; dgtta_argmax_dice routes to it for back-to-back rows)
    against a sequential first-maximum scan: ties (duplicated maxima in different quarters of the class range), -inf rows,
    NaNs anywhere but class 0 (passed over, as the strict comparison of the one-thread-per-row kernel does), ragged tail."""
    from dg_tta_amd import ops
    g = torch.Generator().manual_seed(C * 1000 + rows)
    x = torch.randn(rows, C, generator=g)
    x = (x * 4).round() / 4                                     # many exact ties
    x[3] = float("-inf")
    x[5, :] = 1.0
    if C > 4:
        x[7, C // 2] = float("nan")
        x[8, C - 1] = float("nan")
        x[9, (C + 3) // 4] = float("nan")                       # first class of the second quarter
        x[9, (C + 3) // 4 + 1] = 100.0
    x = x.to(acc_dtype)
    ref = torch.zeros(rows, dtype=torch.int64)
    xf = x.float()
    bv = xf[:, 0].clone()
    for c in range(1, C):
        take = xf[:, c] > bv
        bv = torch.where(take, xf[:, c], bv)
        ref = torch.where(take, torch.full_like(ref, c), ref)
    out = ops.argmax_rows(x.to(DEV))
    assert out.dtype == torch.int64 and torch.equal(out.cpu(), ref)
    if acc_dtype == torch.float32 and rows >= 4096:             # the routed entry point
        am, _ = ops.argmax_dice(x.to(DEV).reshape(1, rows, 1, 1, C).permute(0, 4, 1, 2, 3))
        assert torch.equal(am.reshape(-1).cpu(), ref)


@pytest.mark.gpu
def test_fp16_window_accumulator_against_fp32():
    """Round 4: DGTTA_WINDOW_ACC=fp16 / acc_dtype=torch.float16 (nnU-Net's storage type for predicted_logits): the same
    windows accumulated in half storage (sum in fp32, rounded once per window) stay within half rounding of the fp32
    accumulator - at most 8 overlapping windows, each rounding <= 2^-11 relative to the running sum - and give the same labels
    wherever the top-2 margin exceeds that bound; export_segmentation reads either storage type."""
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.tta.inference import export_segmentation, predict_sliding_window_return_logits
    from dg_tta_amd.unet import HipPlainConvUNet
    net = he_init_(HipPlainConvUNet(act_dtype=torch.bfloat16), seed=7)
    net.decoder.seg_layers[-1].bias.data.normal_()
    net.register_forward_pre_hook(lambda mod, inp: MIND3D(randn_weighting=0.0).forward(*inp, out_dtype=torch.bfloat16))
    net = net.to(DEV)
    torch.manual_seed(4)
    vol = torch.randn(1, 96, 64, 150)
    patch = [64, 64, 64]
    a32, n32, crop = predict_sliding_window_return_logits(net, vol, patch, acc_dtype=torch.float32)
    a16, n16, _ = predict_sliding_window_return_logits(net, vol, patch, acc_dtype=torch.float16)
    assert a16.dtype == torch.float16 and a32.dtype == torch.float32 and torch.equal(n16, n32)
    err = (a16.float() - a32).abs()
    # every partial sum of a voxel is at most nsum x (the largest |logit| of the volume): <= 8 roundings of 2^-11 of that
    lmax = float((a32.abs() / n32[..., None]).max())
    bound = (8 * 2.0 ** -11 * lmax * n32 + 1e-6)[..., None].expand_as(a32)
    assert bool((err <= bound).all()), float((err / bound).max())
    assert float(err.mean()) < 2.0 ** -11 * float(a32.abs().mean())        # and typically one rounding of the final value
    s32 = export_segmentation(a32, n32, crop, None, None, None)
    s16 = export_segmentation(a16, n16, crop, None, None, None)
    top2 = a32.topk(2, dim=-1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 2 * bound.max(-1).values).cpu().numpy()
    assert (s16[safe] == s32[safe]).all() and (s16 == s32).mean() > 0.995 and safe.mean() > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("acc_dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_head_fused_with_the_window_accumulation_is_bit_identical(dtype, acc_dtype, monkeypatch):
    """Round 3: dgtta_seghead_window_accumulate (the head evaluates straight into the Gaussian window accumulator; a
    window's 105-class logits are never written) against head + dgtta_window_accumulate (DGTTA_FUSE_HEAD_ACCUMULATE=0) on
    the full net, through the product's run_inference with the plan's model-output hook in place.  With the FMA-chain
    kernel (DGTTA_HA_MFMA=0): same chain, same accumulation order -> torch.equal accumulators, weight maps and label maps
    (overlapping windows, ragged last step).  The default kernel evaluates the head on the matrix cores with fp32-exact
    three-term weights: the 32-term sums are associated differently - a few 1e-7 of the logits' magnitude, same labels
    outside float ties (round 4)."""
    from conftest import reload_kernel_switches
    from types import SimpleNamespace
    from dg_tta_amd.mind import mind_hook
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.inference import predict_sliding_window_return_logits, run_inference, _can_fuse_head_accumulate
    from dg_tta_amd.tta.model_utils import get_model_from_network
    from dg_tta_amd.unet import HipPlainConvUNet
    net = he_init_(HipPlainConvUNet(act_dtype=dtype), seed=7)
    net.decoder.seg_layers[-1].bias.data.normal_()
    net.register_forward_pre_hook(mind_hook)
    model = get_model_from_network(net, SimpleNamespace(ModifierFunctions=ModifierFunctions), None).to(DEV)
    torch.manual_seed(2)
    vol = torch.randn(1, 96, 64, 150)
    patch = [64, 64, 64]
    outs = {}
    monkeypatch.setenv("DGTTA_WINDOW_ACC", "fp16" if acc_dtype == torch.float16 else "fp32")
    for flag in ("1", "0", "mfma"):
        monkeypatch.setenv("DGTTA_FUSE_HEAD_ACCUMULATE", "0" if flag == "0" else "1")
        monkeypatch.setenv("DGTTA_HA_MFMA", "1" if flag == "mfma" else "0")
        reload_kernel_switches()
        with torch.no_grad():
            assert _can_fuse_head_accumulate(model) == (flag != "0")
        torch.manual_seed(9)                               # MIND noise of the windows: same draws in both runs
        acc, nsum, crop = predict_sliding_window_return_logits(model, vol, patch)
        torch.manual_seed(9)
        seg = run_inference(vol, model, [model.state_dict()], patch)
        outs[flag] = (acc.clone(), nsum.clone(), seg)
    assert tuple(outs["1"][0].shape) == (96, 64, 150, 105) and outs["1"][0].dtype == acc_dtype
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    assert torch.equal(outs["1"][2], outs["0"][2]) and len(outs["1"][2].unique()) > 10
    a_m, n_m, seg_m = outs["mfma"]
    a_r = outs["0"][0].float()
    assert torch.equal(n_m, outs["0"][1])
    lmax = float((a_r.abs() / n_m[..., None]).max())
    tol = (2e-6 if acc_dtype == torch.float32 else 8 * 2.0 ** -11) * lmax * n_m[..., None] + 1e-6      # fp16 storage: a rounding per window may flip; subnormal halves at the window rim
    assert bool(((a_m.float() - a_r).abs() <= tol).all())
    if acc_dtype == torch.float32:
        assert float((a_m - a_r).abs().max()) > 0        # (it IS the other kernel)
    top2 = a_r.topk(2, dim=-1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 2 * tol[..., 0]).cpu()
    assert torch.equal(seg_m[safe], outs["0"][2][safe]) and (seg_m == outs["0"][2]).float().mean() > 0.999
    # a user's model-output modifier in front of the accumulation switches the fusion off
    class Mods(ModifierFunctions):
        @staticmethod
        def modfify_tta_model_output_fn(pred):
            return pred * 2.0
    other = get_model_from_network(net, SimpleNamespace(ModifierFunctions=Mods), None).to(DEV)
    monkeypatch.setenv("DGTTA_FUSE_HEAD_ACCUMULATE", "1")
    with torch.no_grad():
        assert not _can_fuse_head_accumulate(other)


@pytest.mark.gpu
@pytest.mark.parametrize("dts", ["fp32", "bf16", "fp16"])
def test_feature_window_accumulate_kernel(dts):
    """dgtta_feature_window_accumulate against torch: overlapping windows of one batch, ragged volume, every storage type of
    the features; the accumulated Gaussian sums alongside."""
    from dg_tta_amd import _lib
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    dt, tdt = {"fp32": (0, torch.float32), "bf16": (1, torch.bfloat16), "fp16": (2, torch.float16)}[dts]
    torch.manual_seed(4)
    P, (X, Y, Z) = (8, 12, 20), (13, 19, 37)
    gauss = torch.rand(P, device=DEV) + 0.1
    origins = [(0, 0, 0), (5, 7, 0), (5, 7, 10), (3, 2, 17)]
    z = torch.randn(len(origins), *P, 32, device=DEV).to(tdt)
    facc = torch.randn(X, Y, Z, 32, device=DEV)
    nsum = torch.rand(X, Y, Z, device=DEV)
    ref_f, ref_n = facc.clone(), nsum.clone()
    for k, (sx, sy, sz) in enumerate(origins):
        check(lib.dgtta_feature_window_accumulate(ptr(z[k]), ptr(gauss), ptr(facc), ptr(nsum), 32, *P, X, Y, Z, sx, sy, sz, dt,
                                                  stream_of()), "dgtta_feature_window_accumulate")
        ref_f[sx:sx + P[0], sy:sy + P[1], sz:sz + P[2]] += gauss[..., None] * z[k].float()
        ref_n[sx:sx + P[0], sy:sy + P[1], sz:sz + P[2]] += gauss
    torch.cuda.synchronize()
    assert torch.equal(facc, ref_f) and torch.equal(nsum, ref_n)        # one rounded product, one addition: the same bits
    with pytest.raises(RuntimeError):
        check(lib.dgtta_feature_window_accumulate(ptr(z[0]), ptr(gauss), ptr(facc), ptr(nsum), 32, *P, X, Y, Z, 6, 0, 0, dt, stream_of()), "x")
    with pytest.raises(RuntimeError):
        check(lib.dgtta_feature_window_accumulate(ptr(z[0]), ptr(gauss), ptr(facc), ptr(nsum), 16, *P, X, Y, Z, 0, 0, 0, dt, stream_of()), "x")
    # segments of windows that overlap along the last axis (three windows at z = 3, 9, 14 of a row: up to three on a voxel), with and
    # without the folded InstanceNorm apply, against the same windows one by one: the same bits
    import ctypes as C
    zs = [3, 9, 14]
    y = torch.randn(len(zs), *P, 32, device=DEV).to(tdt)
    mr = torch.stack([torch.randn(len(zs), 32, device=DEV), torch.rand(len(zs), 32, device=DEV) + 0.5], -1).contiguous()
    gamma, beta = torch.randn(32, device=DEV), torch.randn(32, device=DEV)
    for norm in (False, True):
        one_f, one_n = facc.clone(), nsum.clone()
        for k, sz in enumerate(zs):
            if norm:
                check(lib.dgtta_feature_window_accumulate_norm(ptr(y[k]), ptr(mr[k]), ptr(gamma), ptr(beta), 0.01, ptr(gauss), ptr(one_f),
                                                               ptr(one_n), 32, *P, X, Y, Z, 2, 1, sz, dt, stream_of()), "norm")
            else:
                check(lib.dgtta_feature_window_accumulate(ptr(y[k]), ptr(gauss), ptr(one_f), ptr(one_n), 32, *P, X, Y, Z, 2, 1, sz, dt,
                                                          stream_of()), "plain")
        seg_f, seg_n = facc.clone(), nsum.clone()
        cuts = sorted(set(zs + [v + P[2] for v in zs]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            cover = [k for k, sz in enumerate(zs) if sz <= a and b <= sz + P[2]]
            srcs = (C.c_void_p * len(cover))(*[y[k].data_ptr() for k in cover])
            mrs = (C.c_void_p * len(cover))(*[mr[k].data_ptr() for k in cover]) if norm else None
            zoffs = (C.c_int * len(cover))(*[a - zs[k] for k in cover])
            check(lib.dgtta_feature_window_accumulate_multi(srcs, mrs, zoffs, len(cover), ptr(gamma), ptr(beta), 0.01, ptr(gauss), ptr(seg_f),
                                                            ptr(seg_n), 32, *P, b - a, X, Y, Z, 2, 1, a, dt, stream_of()), "multi")
        torch.cuda.synchronize()
        assert max(len([k for k, sz in enumerate(zs) if sz <= a and b <= sz + P[2]]) for a, b in zip(cuts[:-1], cuts[1:])) == 3
        assert torch.equal(seg_f, one_f) and torch.equal(seg_n, one_n) and not torch.equal(seg_f, facc)


@pytest.mark.gpu
@pytest.mark.parametrize("M,C,V", [(1, 105, 64 * 40 + 17), (3, 105, 5000), (2, 118, 777), (4, 7, 300), (1, 2, 255), (5, 150, 513)])
def test_feature_head_argmax_and_logits_chunk(M, C, V, monkeypatch):
    """The head applied to the feature accumulators of M members, fused with the argmax (first maximum wins) - on the fp32 matrix
    cores (default, up to 128 classes) and on the vector ALU (DGTTA_FEATURE_HEAD_MFMA=0) - and the export path's class chunks in
    double, against torch on the same numbers."""
    from conftest import reload_kernel_switches
    from dg_tta_amd import _lib, ops
    from dg_tta_amd._lib import check, ptr, stream_of
    lib = _lib.load()
    torch.manual_seed(M * 1000 + C)
    X, Y, Z = 1, 1, V
    facc = torch.randn(M, X, Y, Z, 32, device=DEV)
    nsum = torch.rand(X, Y, Z, device=DEV) + 0.5
    w = torch.randn(M, C, 32, device=DEV)
    b = torch.randn(M, C, device=DEV)
    facc[:, 0, 0, 5] = 0.0                    # a voxel whose classes tie up to the bias ...
    if C > 3:
        b[:, 3] = b[:, 0]                     # ... and two identical classes: the lower id wins
        w[:, 3] = w[:, 0]
    bsum = b.sum(0)
    ref = (torch.einsum("mvk,mck->vc", facc.reshape(M, V, 32).double(), w.double()) + nsum.reshape(V, 1).double() * bsum.double())
    am = ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4 * ref.abs().amax(1)
    for mfma in ("1", "0"):
        monkeypatch.setenv("DGTTA_FEATURE_HEAD_MFMA", mfma)
        reload_kernel_switches()
        seg = ops.feature_head_argmax(facc, nsum, w, bsum)
        got = seg.reshape(V)
        assert torch.equal(got[safe], am[safe]) and float(safe.float().mean()) > 0.9, mfma
        assert int(got.min()) >= 0 and int(got.max()) < C
        if C > 3:
            assert int((got == 3).sum()) == 0        # classes 0 and 3 are identical: the first one is reported
        # a voxel whose features are all zero and whose weight sum is zero: every logit is exactly 0 -> class 0
        f2, n2 = facc.clone(), nsum.clone()
        f2[:, 0, 0, 7] = 0.0
        n2[0, 0, 7] = 0.0
        assert int(ops.feature_head_argmax(f2, n2, w, bsum).reshape(V)[7]) == 0
    monkeypatch.delenv("DGTTA_FEATURE_HEAD_MFMA")
    reload_kernel_switches()
    # export chunk: normalised ensemble logits in double
    c0, cg = (C // 2, min(8, C - C // 2))
    dst = torch.empty(V, cg, dtype=torch.float64, device=DEV)
    check(lib.dgtta_feature_logits_chunk_f64(ptr(facc), facc.stride(0), ptr(nsum), ptr(w), ptr(bsum), ptr(dst), M, 32, C, X, Y, Z, 0, 0, 0,
                                             X, Y, Z, c0, cg, stream_of()), "dgtta_feature_logits_chunk_f64")
    ref_c = (ref / nsum.reshape(V, 1).double())[:, c0:c0 + cg]
    assert float((dst - ref_c).abs().max()) < 1e-10 * float(ref_c.abs().max()) + 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_feature_space_accumulator_against_logits_space_on_the_full_net(dtype, monkeypatch):
    """Round 5: run_inference through the feature-space accumulator (the default) against the logits-space one
    (DGTTA_WINDOW_ACC=fp32) on the full 3d_fullres net, through the plan's model-output hook, overlapping windows and a ragged
    last step, an ensemble of two members with different heads: the same feature maps enter both (same MIND draws), so the
    accumulated logits differ by the re-association of fp32 sums only, and the label maps agree outside float ties."""
    from types import SimpleNamespace
    from dg_tta_amd.mind import mind_hook
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.tta import inference as pinf
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.model_utils import get_model_from_network
    from dg_tta_amd.unet import HipPlainConvUNet
    net = he_init_(HipPlainConvUNet(act_dtype=dtype), seed=7)
    net.decoder.seg_layers[-1].bias.data.normal_()
    net.register_forward_pre_hook(mind_hook)
    model = get_model_from_network(net, SimpleNamespace(ModifierFunctions=ModifierFunctions), None).to(DEV)
    second = {k: v.clone() for k, v in model.state_dict().items()}
    for k in second:
        if "seg_layers" in k:
            second[k] = second[k] + 0.05 * torch.randn_like(second[k])
    params = [model.state_dict(), second]
    params = [{k: v.clone() for k, v in p.items()} for p in params]
    torch.manual_seed(2)
    vol = torch.randn(1, 96, 64, 150)
    patch = [64, 64, 64]
    with torch.no_grad():
        assert pinf._can_accumulate_features(model)
    monkeypatch.delenv("DGTTA_WINDOW_ACC", raising=False)
    torch.manual_seed(9)
    acc_f, nsum_f, crop = pinf.predict_ensemble(vol, model, params, patch)
    assert isinstance(acc_f, pinf.WindowFeatures) and tuple(acc_f.facc.shape) == (2, 96, 64, 150, 32)
    seg_f = torch.as_tensor(pinf.export_segmentation(acc_f, nsum_f, crop, None, None, None))
    monkeypatch.setenv("DGTTA_WINDOW_ACC", "fp32")
    torch.manual_seed(9)
    acc_l, nsum_l, _ = pinf.predict_ensemble(vol, model, params, patch)
    assert torch.is_tensor(acc_l) and tuple(acc_l.shape) == (96, 64, 150, 105)
    seg_l = torch.as_tensor(pinf.export_segmentation(acc_l, nsum_l, crop, None, None, None))
    assert torch.equal(nsum_f, nsum_l)
    lf = acc_f.logits()
    scale = float(acc_l.abs().max())
    assert float((lf - acc_l).abs().max()) < 3e-5 * scale            # fp32 sums of 32 + 8 terms in another order
    top2 = acc_l.topk(2, dim=-1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-4 * scale).cpu()
    assert torch.equal(seg_f[safe], seg_l[safe]) and float(safe.float().mean()) > 0.5
    assert float((seg_f == seg_l).float().mean()) > 0.9999 and len(seg_f.unique()) > 10
    # run_inference takes the same route
    monkeypatch.delenv("DGTTA_WINDOW_ACC", raising=False)
    torch.manual_seed(9)
    assert torch.equal(pinf.run_inference(vol, model, params, patch), seg_f.long())
    # round 6: the window stack and the MIND descriptor of the next batch are produced on a side stream beside the network pass
    # of the current one (default here: mind_hook behind the untouched template modifier); in line (DGTTA_INFER_PREFETCH=0) the
    # same kernels see the same noise draws in the same order: the same bits, in both accumulators
    assert pinf._mind_ahead_ok(model, DEV)
    monkeypatch.setenv("DGTTA_INFER_PREFETCH", "0")
    assert not pinf._mind_ahead_ok(model, DEV)
    torch.manual_seed(9)
    acc_i, nsum_i, _ = pinf.predict_ensemble(vol, model, params, patch)
    assert torch.equal(acc_i.facc, acc_f.facc) and torch.equal(nsum_i, nsum_f)
    monkeypatch.setenv("DGTTA_WINDOW_ACC", "fp32")
    torch.manual_seed(9)
    acc_il, _, _ = pinf.predict_ensemble(vol, model, params, patch)
    assert torch.equal(acc_il, acc_l)
    monkeypatch.delenv("DGTTA_WINDOW_ACC")
    monkeypatch.delenv("DGTTA_INFER_PREFETCH")
    # the InstanceNorm + LeakyReLU apply of the block in front of the head runs inside the accumulation kernel (default) or as its
    # own pass (DGTTA_FEATURE_FOLD=0): the same z values, the same bits in the accumulator
    monkeypatch.setenv("DGTTA_FEATURE_FOLD", "0")
    torch.manual_seed(9)
    acc_u, nsum_u, _ = pinf.predict_ensemble(vol, model, params, patch)
    assert torch.equal(acc_u.facc, acc_f.facc) and torch.equal(nsum_u, nsum_f)
    # ... and the windows of a pass that overlap along the last axis are accumulated segment by segment in one launch each (default) or
    # window by window (DGTTA_FEATURE_SEGMENTS=0): contributions added in the same order - the same bits again, folded or not
    for fold in ("0", "1"):
        monkeypatch.setenv("DGTTA_FEATURE_FOLD", fold)
        monkeypatch.setenv("DGTTA_FEATURE_SEGMENTS", "0")
        torch.manual_seed(9)
        acc_w, nsum_w, _ = pinf.predict_ensemble(vol, model, params, patch)
        assert torch.equal(acc_w.facc, acc_f.facc) and torch.equal(nsum_w, nsum_f)
