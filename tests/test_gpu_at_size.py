"""Parity on the launch shape bench.py times: the FULL 3d_fullres net on 128^3 patches of a 160^3
volume, 2 branches x 4 accumulation steps as ONE batch of 8 through forward AND backward (4-units-per-workgroup weight
gradient sweep, side streams, MIND precomputed on the input stream), GIN + affine in both branches, C_opt = 16.

* one network pass: fp32 MFMA, fp16 and bf16 storage against the fp32 VALU kernels (conv_impl = 1) on the same draws:
  logits, consistency loss, every parameter gradient (cosine / sign agreement per tensor);
* N adaptation epochs of the product's tta_epoch: per-epoch loss delta, pseudo-Dice delta, final label agreement and
  hard-Dice delta of the 16-bit storage types against fp32 - north_star's tolerance (1e-3) is asserted for both;
* ONE accumulation step at 128^3, forward and backward, against the CPU ORACLE (round 4).
Measured numbers are written to gpurun_out/at_size_parity.json (quoted in DESIGN.md)."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
ROOT = Path(__file__).resolve().parents[1]
TOL = 1e-3


def _bench():
    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    import bench
    return bench


def _runner(dtype, impl=0):
    bench = _bench()
    # (the recipe these tests' limits were measured on: GIN + MIND pre-training, the standard target case, lr 3e-4 - bench.py's own
    # default workload is the regime in which the adaptation helps, round 6)
    args = bench.parse_args(["--impl", str(impl), "--pretrain-hooks", "GIN_MIND", "--target-noise", "0.12", "--lr", "3e-4"])
    return bench.EpochRunner(args, DEV, 0, dtype)


def _record(key, value):
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    f = out / "at_size_parity.json"
    d = json.loads(f.read_text()) if f.exists() else {}
    d[key] = value
    f.write_text(json.dumps(d, indent=1))


def _one_pass(runner, seed, steps=4):
    """2 branches x `steps` accumulation steps as one batch through the product's own prepare / run functions, loss and
    backward.  Returns (logits of both branches in the common frame [2*steps,C,D,H,W], loss, dice, unscaled gradients)."""
    from dg_tta_amd import ops
    from dg_tta_amd.gin import gin_aug
    from dg_tta_amd.tta.torch_utils import get_batch
    from dg_tta_amd.tta.tta import prepare_both_branches, run_both_branches
    torch.manual_seed(seed)
    np.random.seed(seed)
    model, cfg = runner.model, runner.cfg
    model.train()

    def next_imgs():
        with torch.no_grad():
            imgs, _ = get_batch(runner.data, [0], runner.patch, fixed_patch_idx=None, device=DEV)
        return imgs[0]

    prepared = prepare_both_branches(cfg, model, gin_aug, 1, next_imgs, DEV, steps=steps, precompute_mind=True)
    ta, tb = run_both_branches(prepared, cfg, model, runner.mapping, cfg["optimized_labels"], runner.modmod, runner.fused,
                               steps=steps)
    loss, dice = ops.consistency_loss(ta, tb, 1)
    scale = float(runner.opt.grad_scale)
    torch.autograd.backward(loss, grad_tensors=torch.full((), scale, dtype=torch.float32, device=DEV))
    torch.cuda.synchronize()
    grads = {n: (p.grad.detach().clone() / scale) for n, p in model.named_parameters() if p.grad is not None}
    logits = ta._dgtta_pair.detach()
    runner.opt.zero_grad()
    return logits, float(loss), dice.detach().clone(), grads


def _grad_stats(g, ref):
    """Per-tensor cosine and sign agreement (over elements that matter: |ref| above 1e-3 of the tensor's max)."""
    out = {}
    for n, r in ref.items():
        if n not in g or r.abs().max() == 0:
            continue
        a, b = g[n].double().flatten(), r.double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))
        big = b.abs() > 1e-3 * b.abs().max()
        sign = float((torch.sign(a[big]) == torch.sign(b[big])).double().mean())
        out[n] = (cos, sign)
    return out


def test_one_pass_batch8_128_every_storage_type_vs_fp32_valu():
    t0 = time.perf_counter()
    ref_r = _runner("fp32", impl=1)
    ref_logits, ref_loss, ref_dice, ref_grads = _one_pass(ref_r, 77)
    t_ref = time.perf_counter() - t0
    del ref_r
    torch.cuda.empty_cache()
    rng = float(ref_logits.max() - ref_logits.min())
    assert len(ref_grads) > 60 and ref_logits.shape == (8, 16, 128, 128, 128)
    report = {"fp32_valu_seconds": round(t_ref, 1), "loss_fp32_valu": ref_loss, "logit_range": rng}
    # bias gradients in front of InstanceNorm are exact zeros in the product setting: excluded by _grad_stats (max == 0)
    # measured (profiles/r03_at_size_parity.json): fp32 MFMA 3.4e-6 / 6e-8 / 0.99997 / 0.992, fp16 1.6e-3 / 3e-7 / 0.987 /
    # 0.943, bf16 1.3e-2 / 1.5e-5 / 0.901 / 0.827 (logit error over range / loss delta / min gradient cosine / min sign agreement)
    # the limits are 2-3x those measurements (the old ones would not have caught a 10x regression)
    # Round 5: the runner's weights are PRE-TRAINED (bench.pretrained_weights) - decisive logits, gradients far above rounding
    # noise.  Measured on them (profiles/r05_at_size_parity.json "one_pass"): fp32 MFMA 7.7e-7 / 6e-8 / 0.9999995 / 0.999998, fp16
    # 2.1e-4 / 3e-7 / 0.9989 / 0.991, bf16 1.8e-3 / 8e-5 / 0.9996 / 0.969, labels 0.9999999 / 0.99995 / 0.9996
    # (550 pre-training steps: fp32 9.7e-7 / 6e-8 / 0.9999997 / 0.999994, fp16 2.8e-4 / 3.5e-6 / 0.9988 / 0.9969, labels 0.99998;
    # the same 550 steps after the weight-gradient kernel changed its summation order - other weights, the pre-training runs
    # through it: fp32 9.9e-7 / 0 / 0.9999916 / 0.999976, fp16 2.9e-4 / 2.1e-5 / 0.99976 / 0.99955.  The fp16 loss delta is a
    # SIGNED sum of rounding errors - 3e-7, 3.5e-6, 2.1e-5 on three sets of weights - and so is the worst tensor's cosine:
    # their limits leave room for that spread; the logit, label and median limits are the tight ones)
    limits = {"fp32": dict(logit=3e-6, loss=3e-7, cos=0.99997, sign=0.9999, agree=0.999999),
              "fp16": dict(logit=1e-3, loss=1e-4, cos=0.996, sign=0.97, agree=0.9998),
              "bf16": dict(logit=6e-3, loss=3e-4, cos=0.997, sign=0.92, agree=0.999)}
    for dtype in ("fp32", "fp16", "bf16"):
        r = _runner(dtype, impl=0)
        logits, loss, dice, grads = _one_pass(r, 77)
        del r
        err = float((logits - ref_logits).abs().max()) / rng
        stats = _grad_stats(grads, ref_grads)
        cos_min = min(c for c, _ in stats.values())
        sign_min = min(s for _, s in stats.values())
        worst = min(stats, key=lambda k: stats[k][0])
        agree = float((logits.argmax(1) == ref_logits.argmax(1)).float().mean())
        report[dtype] = {"logit_err_over_range": err, "loss": loss, "loss_delta": abs(loss - ref_loss),
                         "soft_dice_delta_max": float((dice - ref_dice).abs().max()),
                         "grad_cosine_min": cos_min, "grad_cosine_worst_tensor": worst, "grad_sign_agreement_min": sign_min,
                         "grad_cosine_median": float(np.median([c for c, _ in stats.values()])),
                         "label_agreement": agree, "tensors": len(stats)}
        _record("one_pass", report)
        lim = limits[dtype]
        assert err < lim["logit"], f"{dtype}: logits off by {err:.2e} of their range"
        assert abs(loss - ref_loss) < lim["loss"], f"{dtype}: loss {loss:.6f} vs {ref_loss:.6f}"
        assert cos_min > lim["cos"], f"{dtype}: gradient cosine {cos_min:.4f} in {worst}"
        assert sign_min > lim["sign"], f"{dtype}: gradient sign agreement {sign_min:.4f}"
        assert agree >= lim["agree"], f"{dtype}: label agreement {agree:.7f}"
        del logits, grads
        torch.cuda.empty_cache()


def test_one_step_128_backward_vs_cpu_oracle():
    """BASELINE-size parity against the ORACLE, forward AND backward.  One accumulation step at 128^3 (2 branches:
    GIN -> affine warp -> MIND -> full 3d_fullres net -> inverse warp; masked soft-Dice; backward), the oracle run once on
    the host (oracle/tta.py on torch CPU, the step bench.py's cpu_baseline times), the HIP path in fp32, fp16 and bf16 storage
    on the same image, draws and weights: loss, soft Dice per class, label maps (everywhere, and where the oracle's top-2
    margin exceeds 1e-3), every parameter gradient (cosine / sign agreement per tensor)."""
    bench = _bench()
    import os
    t0 = time.perf_counter()
    dt, rec = bench.oracle_step(128, 16, 16, threads=min(16, len(os.sched_getaffinity(0))))
    report = {"oracle_seconds": round(time.perf_counter() - t0, 1), "oracle_loss": rec["loss"]}
    # measured (profiles/r04_at_size_parity.json, "oracle_step"): the limits are 2-3x those values
    #   fp32: loss 0 .. 6e-8, soft Dice 1.5e-6, logits 4e-5 of their range, labels 0.99996 (1.0 where the margin > 1e-3), min gradient
    #         cosine 0.9998 / sign agreement 0.994;   fp16: 9e-7, 2.3e-5, 2.7e-3, 0.9978 (0.9987), 0.987 / 0.951;
    #   bf16: 2.3e-5, 1.1e-4, 1.7e-2, 0.983 (0.983), 0.897 / 0.865
    limits = {"fp32": dict(loss=5e-7, dice=5e-6, logit=1.2e-4, agree=0.99988, agree_safe=1.0, cos=0.9994, sign=0.98),
              "fp16": dict(loss=3e-6, dice=7e-5, logit=7e-3, agree=0.9945, agree_safe=0.996, cos=0.965, sign=0.88),
              "bf16": dict(loss=7e-5, dice=3.5e-4, logit=4.5e-2, agree=0.955, agree_safe=0.955, cos=0.75, sign=0.68)}
    for dtype in ("fp32", "fp16", "bf16"):
        report[dtype] = r = bench.hip_step_vs_oracle(rec, dtype, DEV)
        _record("oracle_step", report)
        lim = limits[dtype]
        assert r["loss_delta"] < lim["loss"], f"{dtype}: loss {r['loss']:.7f} vs oracle {rec['loss']:.7f}"
        assert r["soft_dice_per_class_delta_max"] < lim["dice"], f"{dtype}: soft Dice off by {r['soft_dice_per_class_delta_max']:.2e}"
        assert r["logit_err_over_range"] < lim["logit"], f"{dtype}: logits off by {r['logit_err_over_range']:.2e} of their range"
        assert r["argmax_agreement"] >= lim["agree"], f"{dtype}: label agreement {r['argmax_agreement']:.6f}"
        assert r["argmax_agreement_where_margin_gt_1e-3"] >= lim["agree_safe"], f"{dtype}: {r['argmax_agreement_where_margin_gt_1e-3']:.6f}"
        assert r["grad_cosine_min"] > lim["cos"], f"{dtype}: gradient cosine {r['grad_cosine_min']:.5f} in {r['grad_cosine_worst_tensor']}"
        assert r["grad_sign_agreement_min"] > lim["sign"], f"{dtype}: gradient sign agreement {r['grad_sign_agreement_min']:.4f}"
    assert report["fp32"]["argmax_agreement_where_margin_gt_1e-3"] == 1.0      # north_star: identical label maps in fp32 (off ties)


def test_one_step_128_on_pretrained_weights_vs_cpu_oracle():
    """the same one-step comparison on the PRE-TRAINED weights and the target-domain patch the bench runs on
    (the He-initialised net above has pseudo-Dice ~0.005 and near-tied logits: bf16's per-class soft-Dice error reads 3.5e-4 there
    and 4.8e-3 on a model that segments).  GIN + MIND pre-training through the engine (550 steps, ~23 s), the centre patch of the
    standard target case, the CPU oracle's accumulation step on the host, the HIP path in all three storage types.  Measured
    (profiles/r05_bench_lines.json `parity_at_size`): fp32 loss 2e-7 / soft Dice 2.3e-6 / labels 1.0 where the margin > 1e-3 /
    gradient cosine 0.99998; fp16 5.9e-5 / 4.7e-4 / 0.999998 / 0.96 (one near-cancelling bias; median 0.99999); bf16 4.4e-4 /
    4.8e-3 / 0.99987 / 0.9992 on the bench's draw pair.  fp32 and fp16 are held to the stated loss tolerances; bf16 is fenced, not claimed."""
    bench = _bench()
    import os
    from dg_tta_amd.synthetic import atlas_case
    args = bench.parse_args(["--pretrain-hooks", "GIN_MIND", "--target-noise", "0.12", "--lr", "3e-4"])
    state, rep = bench.pretrained_weights(args, DEV)
    assert rep["hard_dice_unseen_source_case"] > 0.85
    o = (bench.volume_edge(128) - 128) // 2
    img = atlas_case(bench.volume_edge(128), 15, 31, "target", noise=0.12)[0][None, None, o:o + 128, o:o + 128, o:o + 128].contiguous()
    dt, rec = bench.oracle_step(128, 16, 16, threads=min(16, len(os.sched_getaffinity(0))), state=state, imgs=img)
    assert 0.02 < rec["loss"] < 0.9                      # the consistency mask is alive on this model
    # measured on this draw pair (seeds 101 / 102; gpurun_out/at_size_parity.json "oracle_step_pretrained"): fp32 loss 1.4e-6, per-class
    # soft Dice 2.1e-5, labels 1.0, cosine 0.99999; fp16 1.7e-4 / 1.08e-3 / 0.999997 / 0.976.  The STATED tolerance is on the loss
    # (1e-5 fp32, 1e-3 16-bit - one step is noisier than an epoch's mean of 16); the per-class soft Dice of ONE
    # augmentation pair (small classes through the reference's hard mask) is fenced at ~3x its measured value, not claimed
    limits = {"fp32": dict(loss=1e-5, dice=1e-4, agree_safe=1.0, cos=0.9995),
              "fp16": dict(loss=1e-3, dice=3e-3, agree_safe=0.99995, cos=0.9),
              "bf16": dict(loss=3e-3, dice=3e-2, agree_safe=0.9995, cos=0.99)}
    report = {"oracle_loss": rec["loss"], "pretraining": {k: rep[k] for k in ("steps", "hard_dice_unseen_source_case")}}
    for dtype in ("fp32", "fp16", "bf16"):
        report[dtype] = r = bench.hip_step_vs_oracle(rec, dtype, DEV)
        _record("oracle_step_pretrained", report)
        lim = limits[dtype]
        assert r["loss_delta"] < lim["loss"], f"{dtype}: loss {r['loss']:.7f} vs oracle {rec['loss']:.7f}"
        assert r["soft_dice_per_class_delta_max"] < lim["dice"], f"{dtype}: soft Dice off by {r['soft_dice_per_class_delta_max']:.2e}"
        assert r["argmax_agreement_where_margin_gt_1e-3"] >= lim["agree_safe"], f"{dtype}: {r['argmax_agreement_where_margin_gt_1e-3']:.6f}"
        assert r["grad_cosine_min"] > lim["cos"], f"{dtype}: gradient cosine {r['grad_cosine_min']:.5f} in {r['grad_cosine_worst_tensor']}"


def test_adaptation_epochs_128_dice_delta_of_the_default_storage_type():
    """N = 4 adaptation epochs of the product's tta_epoch (default 2 x 4 batching, side streams), same seeds and draws for
    fp32, fp16 and bf16 storage: north_star's tolerance holds for the 16-bit storage type bench.py defaults to (fp16 since
    round 6) and for BASELINE config 2's bf16; `dgtta run_tta` itself defaults to fp32, the reference's precision."""
    bench = _bench()
    from dg_tta_amd.run import DEFAULT_DTYPE, FAST_DTYPE
    assert DEFAULT_DTYPE == "fp32" and bench.parse_args([]).dtype == FAST_DTYPE
    epochs = 4
    legs = {}
    for dtype in ("fp32", "fp16", "bf16"):
        torch.manual_seed(4321)
        np.random.seed(4321)
        r = _runner(dtype)
        for _ in range(epochs):
            r.epoch()
        labels, per_class = r.final_labels()
        legs[dtype] = (list(r.losses), list(r.dices), labels, per_class, int(r.opt.skipped_steps))
        del r
        torch.cuda.empty_cache()
    ref = legs["fp32"]
    report = {"epochs": epochs, "loss_fp32": ref[0], "pseudo_dice_fp32": ref[1]}
    for dtype in ("fp16", "bf16"):
        losses, dices, labels, per_class, skipped = legs[dtype]
        pc = (per_class - ref[3]).abs()
        pc = pc[~torch.isnan(pc)]
        report[dtype] = {"loss_delta_per_epoch": [abs(a - b) for a, b in zip(losses, ref[0])],
                         "pseudo_dice_delta_per_epoch": [abs(a - b) for a, b in zip(dices, ref[1])],
                         "hard_dice_per_class_delta_max": float(pc.max()),
                         "hard_dice_mean_delta": abs(float(per_class.nanmean()) - float(ref[3].nanmean())),
                         "label_agreement": float((labels == ref[2]).float().mean()), "skipped_steps": skipped}
    _record("epochs", report)
    # Round 5: the runner starts from weights PRE-TRAINED on the source domain of the synthetic atlas task (bench.pretrained_weights)
    # and adapts to a target-domain case at lr 3e-4: the hard Dice vs ground truth is ~0.68 before and moves by ~0.05 in 4 epochs,
    # so north_star's 1e-3 is a real bound here (rounds 3-4 measured it on He-initialised weights at Dice 0.005).  Measured (bench
    # line of profiles/r05_bench_lines.json, 6 epochs): loss 2.7e-5 / 1.6e-4, pseudo-Dice 2.2e-4 / 3.4e-4, hard Dice mean 2.2e-4 /
    # 1.3e-4, labels 0.99972 / 0.99887 (fp16 / bf16)
    assert float(ref[3].nanmean()) > 0.5
    for dtype, lab in (("fp16", 0.999), ("bf16", 0.997)):
        d = report[dtype]
        assert max(d["loss_delta_per_epoch"]) < (3e-4 if dtype == "fp16" else 5e-4) < TOL      # (550 steps: 1.0e-4 / 3.9e-5)
        assert max(d["pseudo_dice_delta_per_epoch"]) < TOL
        assert d["hard_dice_mean_delta"] < TOL
        assert d["skipped_steps"] == 0
        assert d["label_agreement"] > lab


def _volume_512(k=15):
    """BASELINE config 3's input: a 512^3 "CT" with k label channels, [1+k, 512, 512, 512] fp32 on the host (8.6 GB)."""
    g = torch.Generator().manual_seed(33)
    n = 512
    low = torch.randn(1, 1, 34, 34, 34, generator=g)
    img = torch.nn.functional.interpolate(low, size=(n, n, n), mode="trilinear", align_corners=False)[0, 0]
    data = torch.empty((1 + k, n, n, n), dtype=torch.float32)
    data[0] = img
    # labels: 64^3 blocks numbered 0..k in a fixed pattern (every label present in every 128^3 patch neighbourhood)
    ax = torch.arange(n) // 64
    lab = ((ax[:, None, None] * 5 + ax[None, :, None] * 3 + ax[None, None, :]) % (k + 1)).to(torch.int16)
    for i in range(k):
        data[1 + i] = (lab == i + 1)
    return data, lab


def test_config3_tta_epoch_from_a_512_cubed_resident_volume():
    """BASELINE config 3 at size, TTA side: patches sampled from a 512^3 volume with 15 label channels that stays
    resident in HBM (0.5 GB image + label map), one adaptation epoch of the product's tta_epoch (full net, 128^3 patches,
    16 accumulation steps) and its evaluation against the label channels."""
    bench = _bench()
    from dg_tta_amd.tta.torch_utils import _VOLUME_CACHE, get_batch, release_resident
    release_resident()
    data, lab = _volume_512()
    r = _runner(bench.parse_args([]).dtype)
    r.data = [data]
    torch.manual_seed(5)
    np.random.seed(5)
    # the sampler: label patches are bit-exact crops for the centre patch (nearest sampling of an aligned grid)
    imgs, labels = get_batch(r.data, [0], r.patch, fixed_patch_idx="center", device=DEV)
    assert tuple(imgs[0].shape) == (1, 1, 128, 128, 128) and labels[0].dtype == torch.int64
    assert torch.equal(labels[0][0, 0].cpu(), lab[192:320, 192:320, 192:320].long())
    assert torch.allclose(imgs[0][0, 0].cpu(), data[0, 192:320, 192:320, 192:320], atol=1e-5)
    assert len(_VOLUME_CACHE) == 1
    before = {n: p.detach().clone() for n, p in list(r.model.named_parameters())[:4]}
    r.epoch()
    r.epoch()
    assert np.isfinite(r.losses).all() and 0.0 < r.losses[-1] < 1.0 and np.isfinite(r.dices).all()
    assert any(not torch.equal(before[n], p.detach()) for n, p in list(r.model.named_parameters())[:4])
    assert len(_VOLUME_CACHE) == 1               # uploaded once, sampled 2 x (16 + 1) times
    release_resident(r.data)
    assert len(_VOLUME_CACHE) == 0


def test_config3_sliding_window_512_cubed_properties():
    """BASELINE config 3 at size, inference side: 512^3 volume, 128^3 Gaussian windows at step 0.5 -> 7^3 = 343 windows, all
    105 classes accumulated in fp32 (52.5 GiB).  Size-independent properties: the accumulated weight map is the sum of the
    window Gaussians; a region covered by ONE window equals the plain forward of that window; the label map is the argmax
    over all 105 classes of the accumulator."""
    from dg_tta_amd import ops
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.run import FAST_DTYPE as DEFAULT_DTYPE
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.tta.inference import (compute_gaussian, compute_steps_for_sliding_window, export_segmentation,
                                          predict_sliding_window_return_logits)
    from dg_tta_amd.unet import HipPlainConvUNet
    adt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[DEFAULT_DTYPE]
    net = he_init_(HipPlainConvUNet(act_dtype=adt), seed=7).to(DEV)
    net.register_forward_pre_hook(lambda mod, inp: MIND3D(randn_weighting=0.0).forward(*inp, out_dtype=adt))
    patch = [128, 128, 128]
    vol = torch.randn(1, 512, 512, 512, generator=torch.Generator().manual_seed(3)).to(DEV)
    acc, nsum, crop = predict_sliding_window_return_logits(net, vol, patch)
    assert tuple(acc.shape) == (512, 512, 512, 105) and acc.dtype == torch.float32
    steps = compute_steps_for_sliding_window((512, 512, 512), patch)
    assert steps == [[0, 64, 128, 192, 256, 320, 384]] * 3
    # (a) weight map: separable check - the Gaussian is a product of 1-D profiles only approximately (it is a 3-D filter of
    # a delta), so the reference map is accumulated on the GPU from the same window origins
    g = compute_gaussian(tuple(patch)).to(DEV)
    ref_n = torch.zeros((512, 512, 512), device=DEV)
    for sx in steps[0]:
        for sy in steps[1]:
            for sz in steps[2]:
                ref_n[sx:sx + 128, sy:sy + 128, sz:sz + 128] += g
    assert float((nsum - ref_n).abs().max()) < 1e-4 * float(ref_n.max())
    del ref_n
    # (b) the corner [0:64]^3 is covered by the first window only
    with torch.no_grad():
        first = net(vol[None, :, :128, :128, :128]).float()[0]
    corner = (acc[:64, :64, :64] / nsum[:64, :64, :64, None]).permute(3, 0, 1, 2)
    assert float((corner - first[:, :64, :64, :64]).abs().max()) < 1.5e-2 * float(first.abs().max())
    del corner, first
    # (c) label map = argmax over all 105 classes (checked on a slab: the full top-2 of 52 GiB is not needed for the point)
    seg = export_segmentation(acc, nsum, crop, None, None, None)
    assert seg.shape == (512, 512, 512)
    slab = acc[200:232]
    am = slab.argmax(-1).cpu()
    top2 = slab.topk(2, dim=-1).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-3 * nsum[200:232]).cpu()
    got = torch.from_numpy(seg[200:232].astype(np.int64))
    assert torch.equal(got[safe], am[safe]) and float((got == am).float().mean()) > 0.999
    assert len(np.unique(seg[::16, ::16, ::16])) > 20


def test_config4_eight_volumes_eight_logical_ranks_match_the_sequential_run(tmp_path):
    """BASELINE config 4 as far as one GPU allows (VERDICT r3 #7a): 8 independent synthetic volumes (seeds 0..7), the FULL
    3d_fullres net on 64^3 patches, sample-sharded over 8 logical ranks that share cuda:0 (run one after the other, in an
    order in which rank 0 comes last, on one run directory): every <case>__ensemble_idx_0_tta_parameters.pt is torch.equal to
    the one of the sequential single-process run, every case is predicted exactly once, the summary is written once (by rank
    0, after all eight done-markers) and lists the eight cases.  What stays untested here is eight PHYSICAL GPUs."""
    import json
    from types import SimpleNamespace as NS
    from dg_tta_amd.gin import gin_hook
    from dg_tta_amd.mind import mind_hook
    from dg_tta_amd.synthetic import he_init_, synthetic_case, synthetic_label_mapping
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions, TEMPLATE_PLAN
    from dg_tta_amd.tta.tta import tta_main
    from dg_tta_amd.unet import HipPlainConvUNet
    net = he_init_(HipPlainConvUNet(act_dtype=torch.bfloat16), seed=7)
    net.register_forward_pre_hook(gin_hook)
    net.register_forward_pre_hook(mind_hook)
    net = net.to(DEV)
    mapping, names = synthetic_label_mapping(7)
    cfg = dict(TEMPLATE_PLAN)
    cfg.update(do_intensity_aug_in="both", do_spatial_aug_in="both", patches_to_be_accumulated=4, lr=1e-5, epochs=3,
               ensemble_count=1, optimized_labels=names, tta_data_filepaths=[], seed=11, pretrained_weights_filepath="unused",
               barrier_timeout_s=2.0)
    modmod = NS(ModifierFunctions=ModifierFunctions)
    bundle = (NS(), [64, 64, 64], net, [{k: v.clone() for k, v in net.state_dict().items()}])

    def data():
        return iter([{"data": synthetic_case(size=80, k=7, seed=s), "data_properties": {}, "ofile": f"tta_outputTs/vol{s}"}
                     for s in range(8)]), 8

    seq = tta_main("seq", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data())
    assert ("summary", "Ts") in seq
    results = {}
    for rank in (4, 5, 6, 7, 1, 2, 3, 0):
        results[rank] = tta_main("par", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data(),
                                 shard=(rank, 8))
        assert (("summary", "Ts") in results[rank]) == (rank == 0)
    for s in range(8):
        a = torch.load(tmp_path / "seq" / "tta_outputTs" / f"vol{s}__ensemble_idx_0_tta_parameters.pt", map_location="cpu")[0]
        b = torch.load(tmp_path / "par" / "tta_outputTs" / f"vol{s}__ensemble_idx_0_tta_parameters.pt", map_location="cpu")[0]
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a), f"vol{s}: adapted parameters differ"
        assert np.array_equal(np.load(tmp_path / "seq" / "tta_outputTs" / f"vol{s}.npy"),
                              np.load(tmp_path / "par" / "tta_outputTs" / f"vol{s}.npy"))
        owners = [r for r in range(8) if (f"tta_outputTs/vol{s}", "prediction") in results[r]]
        assert owners == [s]                                        # round robin: sample s on rank s, predicted there, once
    sj = json.loads((tmp_path / "par" / "summary_Ts.json").read_text())
    assert sorted(Path(c["prediction_file"]).name for c in sj["metric_per_case"]) == [f"vol{s}.npy" for s in range(8)]
    assert sj["mean"] == json.loads((tmp_path / "seq" / "summary_Ts.json").read_text())["mean"]


def test_config5_multires_degraded_volumes_fp16_vs_fp32():
    """BASELINE config 5 on one GPU (VERDICT r3 #7b): the GIN_MIND_MultiRes setting differs from config 2 only in what the
    volumes look like - pre-degraded by the discrete zoom factors 1/2, 1/4, 1/6 of dg_tta/pretraining/discrete_downsampling.py:8-37
    (down with order 1, back up with order 0: blocky 3 / 6 / 9 mm data) - and runs fp16 mixed precision.  For each factor: 2
    adaptation epochs of the product's tta_epoch at 128^3 in fp16 storage against fp32 on the same seeds and draws: every Dice
    quantity within north_star's 1e-3, no optimizer step skipped by the loss-scale guard."""
    from dg_tta_amd.pretraining.discrete_downsampling import augment_discrete_linear_downsampling
    report = {}
    for zoom in (1 / 2, 1 / 4, 1 / 6):
        legs = {}
        for dtype in ("fp32", "fp16"):
            torch.manual_seed(99)
            np.random.seed(99)
            r = _runner(dtype)
            vol = r.data[0]
            img = augment_discrete_linear_downsampling(vol[:1].to(DEV), zoom_range=[zoom], p=1.0)       # image channel only
            assert float((img.cpu() - vol[:1]).abs().max()) > 0.1                                      # really degraded
            r.data = [torch.cat([img.cpu(), vol[1:]]).contiguous()]
            torch.manual_seed(4321)
            np.random.seed(4321)
            for _ in range(2):
                r.epoch()
            labels, per_class = r.final_labels()
            legs[dtype] = (list(r.losses), list(r.dices), labels, per_class, int(r.opt.skipped_steps), float(r.opt.grad_scale))
            del r
            torch.cuda.empty_cache()
        ref, got = legs["fp32"], legs["fp16"]
        pc = (got[3] - ref[3]).abs()
        pc = pc[~torch.isnan(pc)]
        report[f"zoom_1/{round(1 / zoom)}"] = ent = {
            "loss_delta_max": max(abs(a - b) for a, b in zip(got[0], ref[0])),
            "pseudo_dice_delta_max": max(abs(a - b) for a, b in zip(got[1], ref[1])),
            "hard_dice_per_class_delta_max": float(pc.max()) if pc.numel() else 0.0,
            "hard_dice_mean_delta": abs(float(got[3].nanmean()) - float(ref[3].nanmean())),
            "label_agreement": float((got[2] == ref[2]).float().mean()), "skipped_steps": got[4], "loss_scale": got[5]}
        _record("config5_multires_fp16", report)
        assert ent["loss_delta_max"] < TOL and ent["pseudo_dice_delta_max"] < TOL and ent["hard_dice_mean_delta"] < TOL
        assert ent["hard_dice_per_class_delta_max"] < TOL
        assert ent["skipped_steps"] == 0 and ent["loss_scale"] == 16384.0
