"""GPU parity of the assembled path: calc_branch and a multi-epoch TTA run against the golden vectors that were
generated with the reference's own calc_branch / soft_dice_loss / torch AdamW (tests/golden/make_golden.py)."""
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden, unpack_draws, SMALL_CFG, state_from_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

LABEL_MAPPING = {"background": (0, 0), "a": (2, 1), "b": (3, 2), "c": (5, 3), "d": (8, 4)}
OPTIMIZED = ["background", "a", "b", "c", "d"]


def _model(g, prefix="w::", **kw):
    from dg_tta_amd.unet import HipPlainConvUNet
    m = HipPlainConvUNet(SMALL_CFG, conv_impl=kw.pop("conv_impl", 1), **kw)
    missing = m.load_state_dict(state_from_golden(g, prefix), strict=False)
    assert not missing.unexpected_keys
    return m.to(DEV)


def hip_branch(model, imgs, draws):
    """calc_branch with explicit draws, HIP ops only."""
    from dg_tta_amd import ops
    from dg_tta_amd.mind import MIND3D
    from oracle import tta as otta
    alpha, ks, kers, shifts = draws["gin_draw"]
    x = ops.gin_chain(imgs, alpha.to(DEV), ks, [k.to(DEV) for k in kers], [s.to(DEV) for s in shifts])
    r, rinv = otta.rand_affine_from_draw(draws["affine_draw"])       # host 4x4 inverse, as augmentation_utils.py:170
    x = ops.affine_warp(x, r.to(DEV), padding_mode="border", tta_grid_algebra=True)
    x = MIND3D()(x, draws["mind_noise"].to(DEV))
    y = model(x)
    return ops.affine_warp(y, rinv.to(DEV), padding_mode="zeros", tta_grid_algebra=True)


@pytest.mark.parametrize("conv_impl", [1, 0])
def test_calc_branch_golden(conv_impl):
    g = load_golden("calc_branch")
    model = _model(g, conv_impl=conv_impl)
    model.set_selected_classes(g["map_idxs"])
    imgs = g["imgs"].to(DEV)
    for br in ("a", "b"):
        out = hip_branch(model, imgs, unpack_draws(g, br))
        ref = g[f"out_{br}"]
        err = (out.cpu() - ref).abs().max().item()
        assert err < 3e-4 * ref.abs().max().item(), f"branch {br}: max err {err:.3e}"
        # label maps: bit-exact argmax wherever the reference's top-2 margin exceeds the float tolerance
        top2 = ref.topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert torch.equal(out.cpu().argmax(1)[safe], ref.argmax(1)[safe])
        assert (out.cpu().argmax(1) == ref.argmax(1)).float().mean() > 0.999


from oracle.replay import cpu_rng_for_device_draws      # noqa: E402  (the reference's one-generator draw stream)


def _plan(**over):
    from dg_tta_amd.tta.config_log_utils import TEMPLATE_PLAN
    cfg = dict(TEMPLATE_PLAN)
    cfg.update(do_intensity_aug_in="both", do_spatial_aug_in="both", patches_to_be_accumulated=2, lr=1e-3,
               optimized_labels=OPTIMIZED)
    cfg.update(over)
    return cfg


def _network_with_hooks(g, **kw):
    from dg_tta_amd.gin import gin_hook
    from dg_tta_amd.mind import mind_hook
    net = _model(g, **kw)
    net.register_forward_pre_hook(gin_hook)      # nnUNetTrainer_GIN_MIND.py:55-57 order
    net.register_forward_pre_hook(mind_hook)
    return net


def _product_model(g, **kw):
    """network with the trainer's pre-hooks + modifier hooks, exactly as tta_main builds it."""
    from dg_tta_amd.gin import gin_hook
    from dg_tta_amd.mind import mind_hook
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.model_utils import get_model_from_network
    from dg_tta_amd.utils import disable_internal_augmentation
    net = _network_with_hooks(g, **kw)
    modmod = SimpleNamespace(ModifierFunctions=ModifierFunctions)
    model = get_model_from_network(net, modmod, None)
    hooks = list(model._forward_pre_hooks.values())
    assert hooks[1] is gin_hook and hooks[2] is mind_hook          # order: modify_input, gin_hook, mind_hook
    disable_internal_augmentation()
    return model, modmod


def test_product_calc_branch_reproduces_reference_draw_order():
    from dg_tta_amd.gin import gin_aug
    from dg_tta_amd.tta.tta import calc_branch, _fuse_head_if_possible
    g = load_golden("calc_branch")
    model, modmod = _product_model(g)
    assert _fuse_head_if_possible(model, modmod, LABEL_MAPPING, OPTIMIZED)
    cfg = _plan()
    imgs = g["imgs"].to(DEV)
    for br, seed in (("a", 51), ("b", 52)):
        with cpu_rng_for_device_draws():
            torch.manual_seed(seed)
            out = calc_branch(f"branch_{br}", cfg, model, gin_aug, None, [16, 16, 16], 1, LABEL_MAPPING, OPTIMIZED,
                              modmod, imgs, torch.device(DEV), head_is_fused=True)
        assert out.requires_grad
        ref = g[f"out_{br}"]
        assert (out.detach().cpu() - ref).abs().max().item() < 3e-4 * ref.abs().max().item()


def test_batched_branches_equal_two_sequential_branches():
    """calc_both_branches (one batch of 2 through the network) == calc_branch a, then b: same draws on both generators
    (the draw order is kept), same targets, same loss and the same accumulated parameter gradients."""
    from dg_tta_amd import ops
    from dg_tta_amd.gin import gin_aug
    from dg_tta_amd.tta.tta import START_CLASS, _fuse_head_if_possible, calc_both_branches, calc_branch
    g = load_golden("calc_branch")
    cfg = _plan()
    imgs = g["imgs"].to(DEV)
    dev = torch.device(DEV)
    results = []
    for batched in (False, True):
        model, modmod = _product_model(g)
        assert _fuse_head_if_possible(model, modmod, LABEL_MAPPING, OPTIMIZED)
        torch.manual_seed(77)
        torch.cuda.manual_seed(78)
        if batched:
            ta, tb = calc_both_branches(cfg, model, gin_aug, [16, 16, 16], 1, LABEL_MAPPING, OPTIMIZED, modmod, imgs, dev,
                                        head_is_fused=True)
        else:
            a = (cfg, model, gin_aug, None, [16, 16, 16], 1, LABEL_MAPPING, OPTIMIZED, modmod, imgs, dev, True)
            ta, tb = calc_branch("branch_a", *a), calc_branch("branch_b", *a)
        loss, _ = ops.consistency_loss(ta, tb, START_CLASS)
        loss.backward()
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        results.append((ta.detach().float().cpu(), tb.detach().float().cpu(), float(loss), grads))
    (ta0, tb0, l0, g0), (ta1, tb1, l1, g1) = results
    assert (ta0 - ta1).abs().max().item() < 1e-5 * max(1.0, ta0.abs().max().item())
    assert (tb0 - tb1).abs().max().item() < 1e-5 * max(1.0, tb0.abs().max().item())
    assert abs(l0 - l1) < 1e-6
    assert g0.keys() == g1.keys() and len(g0) > 10
    for n in g0:
        scale = g0[n].abs().max().item()
        assert (g0[n] - g1[n]).abs().max().item() < 2e-4 * scale + 1e-7, n


def test_batched_steps_equal_sequential_accumulation():
    """2 accumulation steps x 2 branches as ONE batch of 4 == two sequential steps of two sequential branches: same
    draws (patch offsets, GIN, affine, MIND noise on their generators in the reference order), same per-step losses and
    the same accumulated gradients."""
    from dg_tta_amd import ops
    from dg_tta_amd.gin import gin_aug
    from dg_tta_amd.tta.torch_utils import get_batch
    from dg_tta_amd.tta.tta import START_CLASS, _fuse_head_if_possible, calc_both_branches, calc_branch
    g = load_golden("calc_branch")
    cfg = _plan()
    dev = torch.device(DEV)
    vol = torch.randn(1, 24, 24, 24, generator=torch.Generator().manual_seed(5)) * 100.0 - 300.0
    results = []
    for batched in (False, True):
        model, modmod = _product_model(g)
        assert _fuse_head_if_possible(model, modmod, LABEL_MAPPING, OPTIMIZED)
        torch.manual_seed(91)
        torch.cuda.manual_seed(92)

        def next_imgs():
            imgs, _ = get_batch([vol], [0], [16, 16, 16], None, dev)
            return imgs[0]

        losses = []
        if batched:
            ta, tb = calc_both_branches(cfg, model, gin_aug, [16, 16, 16], 1, LABEL_MAPPING, OPTIMIZED, modmod, next_imgs,
                                        dev, head_is_fused=True, steps=2)
            loss, dice = ops.consistency_loss(ta, tb, START_CLASS)
            losses = (1.0 - dice[:, START_CLASS:].mean(1)).tolist()
            assert abs(float(loss) - sum(losses) / 2) < 1e-6
            torch.autograd.backward(loss, grad_tensors=torch.full((), 0.5 * 2, device=DEV))
        else:
            for _ in range(2):
                a = (cfg, model, gin_aug, None, [16, 16, 16], 1, LABEL_MAPPING, OPTIMIZED, modmod, next_imgs(), dev, True)
                ta, tb = calc_branch("branch_a", *a), calc_branch("branch_b", *a)
                loss, _ = ops.consistency_loss(ta, tb, START_CLASS)
                losses.append(float(loss))
                torch.autograd.backward(loss, grad_tensors=torch.full((), 0.5, device=DEV))
        grads = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        results.append((losses, grads))
    (l0, g0), (l1, g1) = results
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 2e-6, (l0, l1)
    assert g0.keys() == g1.keys() and len(g0) > 10
    for n in g0:
        scale = g0[n].abs().max().item()
        assert (g0[n] - g1[n]).abs().max().item() < 3e-4 * scale + 1e-7, n


def test_unfused_head_matches_fused():
    from dg_tta_amd.tta.torch_utils import map_label
    g = load_golden("calc_branch")
    model = _model(g)
    x = torch.randn(1, 12, 16, 16, 16, device=DEV)
    full = model(x)
    assert tuple(full.shape) == (1, 9, 16, 16, 16)
    model.set_selected_classes(g["map_idxs"])
    sel = model(x)
    assert torch.equal(map_label(full, g["map_idxs"], "logits"), sel)


@pytest.mark.parametrize("conv_impl", [1, 0])
def test_tta_epochs_golden(conv_impl):
    """3 epochs x 2 accumulation steps (epoch 0 = loss only) with the golden draws: loss trajectory, updated
    parameters and the final label map."""
    from dg_tta_amd import ops
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.optim import HipAdamW
    from dg_tta_amd.tta.torch_utils import fix_all, release_all
    g = load_golden("tta_epoch")
    model = _model(g, conv_impl=conv_impl)
    model.set_selected_classes(g["map_idxs"])
    opt = HipAdamW(model.parameters(), lr=float(g["lr"]))
    imgs = g["imgs"].to(DEV)
    inv = torch.full((), 0.5, device=DEV)
    losses = []
    model.apply(fix_all)
    for epoch in range(3):
        if epoch == 1:
            model.apply(release_all)
        for acc in range(2):
            ta = hip_branch(model, imgs, unpack_draws(g, f"e{epoch}s{acc}_a"))
            tb = hip_branch(model, imgs, unpack_draws(g, f"e{epoch}s{acc}_b"))
            loss, _ = ops.consistency_loss(ta, tb, 1)
            losses.append(float(loss))
            if epoch >= 1:
                torch.autograd.backward(loss, grad_tensors=inv)
        if epoch >= 1:
            opt.step()
            opt.zero_grad()
    ref_losses = g["losses"].tolist()
    for i, (a, b) in enumerate(zip(losses, ref_losses)):
        # steps 0-3 run on identical weights: 2e-4.  Steps 4-5 follow an AdamW step with lr=1e-3 (100x the plan
        # default, chosen to make the test sensitive); Adam's first step is ~lr*sign(g), so parameters whose gradient
        # is at rounding-noise level move by +-1e-3 with an arbitrary sign on either side: 1e-3 on the loss.
        tol = 2e-4 if i < 4 else 1e-3
        assert abs(a - b) < tol, f"step {i}: loss {a:.6f} vs reference {b:.6f}"
    # parameters after two AdamW steps (lr 1e-3): every update is at most lr per step; compare to the reference's
    post = state_from_golden(g, "p::")
    pre = state_from_golden(g, "w::")
    moved, agree = 0, 0
    for name, p in model.state_dict().items():
        if name not in post:
            continue
        if name.endswith("conv.bias") and ".convs." in name:
            continue            # zero-gradient parameters (bias before InstanceNorm): Adam amplifies rounding noise
        d_ref = post[name] - pre[name]
        d = p.cpu() - pre[name]
        moved += d_ref.numel()
        agree += int(((d - d_ref).abs() <= 0.25 * d_ref.abs() + 2e-5).sum())
    assert agree / moved > 0.93, f"only {agree / moved:.3f} of the parameter updates agree with the reference"
    with torch.no_grad():
        logits = model(MIND3D()(imgs, g["eval_noise"].to(DEV)))
    ref = g["eval_logits"]
    top2 = ref.topk(2, dim=1).values
    # after two sign-like Adam steps with lr=1e-3 the logits agree to ~1e-2 (see the tolerance note above): label
    # maps must be identical wherever the reference's top-2 margin exceeds that, and nearly everywhere overall
    # measured: logits agree to 0.1 max / 0.009 mean on a range of +-5.7, 1 % of the voxels (all with top-2 margin
    # below 0.12) change label.  With the plan's default lr=1e-5 these deviations are 100x smaller.
    assert (logits.cpu() - ref).abs().max() < 0.25 and (logits.cpu() - ref).abs().mean() < 0.02
    safe = (top2[:, 0] - top2[:, 1]) > 0.25
    assert torch.equal(logits.cpu().argmax(1)[safe], g["eval_argmax"][safe])
    assert (logits.cpu().argmax(1) == g["eval_argmax"]).float().mean() > 0.97


def _synthetic_case(seed, size=24, k=3):
    gen = torch.Generator().manual_seed(seed)
    img = torch.randn(1, size, size, size, generator=gen)
    lab = torch.randint(0, k + 1, (size, size, size), generator=gen)
    return torch.cat([img, torch.stack([(lab == i + 1).float() for i in range(k)])])


def test_tta_main_end_to_end(tmp_path):
    """tta_main on two synthetic cases: files, resume-skip, sharding of independent samples."""
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.tta import tta_main
    g = load_golden("calc_branch")
    net = _network_with_hooks(g)
    modmod = SimpleNamespace(ModifierFunctions=ModifierFunctions)
    from types import SimpleNamespace as NS
    cfg = _plan(epochs=2, ensemble_count=2, patches_to_be_accumulated=2, tta_data_filepaths=[], seed=3,
                pretrained_weights_filepath="unused", lr=1e-4)
    mapping = {"background": (0, 0), "a": (2, 1), "b": (3, 2), "c": (5, 3)}
    cfg["optimized_labels"] = ["background", "a", "b", "c"]

    def data():
        return iter([{"data": _synthetic_case(s), "data_properties": {}, "ofile": f"tta_outputTs/case{s}"}
                     for s in (1, 2)]), 2

    params = [{k: v.clone() for k, v in net.state_dict().items()}]
    bundle = (NS(), [16, 16, 16], net, params)
    torch.manual_seed(0)
    np.random.seed(0)
    res = tta_main("run0", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data())
    preds = {k: v for k, v in res.items() if k[1] == "prediction"}
    summ = {k: v for k, v in res.items() if k[0] == "summary"}
    res = {k: v for k, v in res.items() if k[1] != "prediction" and k[0] != "summary"}
    assert len(res) == 4 and len(preds) == 2 and list(summ) == [("summary", "Ts")]
    # evaluation artefacts as the reference writes them: mapped targets + summary_Ts.json in nnU-Net's layout
    import json
    sj = json.loads((tmp_path / "run0" / "summary_Ts.json").read_text())
    assert len(sj["metric_per_case"]) == 2 and set(sj["mean"]) == {"0", "1", "2", "3"}
    assert sj["foreground_mean"]["Dice"] == pytest.approx(summ[("summary", "Ts")], nan_ok=True)
    tgt = np.load(tmp_path / "run0" / "mapped_target_labelsTs" / "case1.npy")
    assert tgt.shape == (24, 24, 24) and set(np.unique(tgt).tolist()) <= {0, 1, 2, 3}
    m1 = sj["metric_per_case"][0]["metrics"]["1"]
    assert m1["TP"] + m1["FN"] == m1["n_ref"] == int((tgt == 1).sum())
    seg = np.load(preds[("tta_outputTs/case1", "prediction")])
    assert seg.shape == (24, 24, 24) and set(np.unique(seg).tolist()) <= {0, 1, 2, 3}
    out = tmp_path / "run0" / "tta_outputTs"
    files = sorted(p.name for p in out.glob("*_tta_parameters.pt"))
    assert files == [f"case{s}__ensemble_idx_{e}_tta_parameters.pt" for s in (1, 2) for e in (0, 1)]
    saved = torch.load(out / files[0], map_location="cpu")
    assert isinstance(saved, list) and set(saved[0].keys()) == set(net.state_dict().keys())
    for (losses, dices) in res.values():
        assert torch.isfinite(losses).all() and (losses > 0).all() and torch.isfinite(dices).all()
    # resume: everything exists -> nothing is recomputed
    again = tta_main("run0", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data())
    assert all(k[1] == "prediction" or k[0] == "summary" for k in again)
    # sharding: rank 1 of 2 owns sample index 1 only, and reproduces the single-process result bit for bit (seeded units)
    res1 = tta_main("run1", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data(),
                    shard=(1, 2))
    res1 = {k: v for k, v in res1.items() if k[1] != "prediction" and k[0] != "summary"}
    assert sorted(k[0] for k in res1) == ["tta_outputTs/case2"] * 2
    a = torch.load(out / "case2__ensemble_idx_1_tta_parameters.pt", map_location="cpu")[0]
    b = torch.load(tmp_path / "run1" / "tta_outputTs" / "case2__ensemble_idx_1_tta_parameters.pt", map_location="cpu")[0]
    for k in a:       # every kernel on the path is deterministic (fixed-order reductions, atomic-free warp backward)
        assert torch.equal(a[k], b[k]), f"sharded run differs from the single-process run in {k}"


@pytest.mark.parametrize("low", ["bf16", "fp16"])
def test_16bit_path_tracks_fp32_within_dice_tolerance(low):
    """bf16 / fp16 storage (fp32 accumulation; fp16 with its static loss scale through HipAdamW) vs the fp32 HIP path on an MFMA-shaped net (32^3): logits, consistency loss,
    label maps, and the loss after adaptation epochs.  north_star tolerance: Dice within 1e-3."""
    from dg_tta_amd import ops
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.optim import HipAdamW
    from dg_tta_amd.synthetic import he_init_
    from dg_tta_amd.unet import HipPlainConvUNet
    from dg_tta_amd.tta.torch_utils import dice_coeff
    from oracle import gin as ogin, tta as otta
    cfg = dict(features=(16, 32, 64), strides=(1, 2, 2), n_conv_enc=(2, 2, 2), n_conv_dec=(2, 2), in_channels=12,
               num_classes=20)
    sel = torch.tensor([0, 3, 5, 7, 11, 13, 17, 19])
    nets = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16 if low == "bf16" else torch.float16)):
        m = he_init_(HipPlainConvUNet(cfg, act_dtype=dt), seed=5).to(DEV)
        m.set_selected_classes(sel)
        nets[name] = m
    torch.manual_seed(0)
    imgs = (torch.randn(1, 1, 32, 32, 32) * 2).to(DEV)

    def draws(seed):
        torch.manual_seed(seed)
        return dict(gin_draw=ogin.draw_gin_params(1), affine_draw=torch.randn(1, 3, 4),
                    mind_noise=torch.randn(1, 12, 32, 32, 32))

    def branch(model, d):
        alpha, ks, kers, shifts = d["gin_draw"]
        x = ops.gin_chain(imgs, alpha.to(DEV), ks, [k.to(DEV) for k in kers], [s.to(DEV) for s in shifts])
        r, rinv = otta.rand_affine_from_draw(d["affine_draw"])
        x = ops.affine_warp(x, r.to(DEV), padding_mode="border", tta_grid_algebra=True)
        x = MIND3D()(x, d["mind_noise"].to(DEV), out_dtype=model.act_dtype)
        return ops.affine_warp(model(x), rinv.to(DEV), padding_mode="zeros", tta_grid_algebra=True)

    hist = {k: [] for k in nets}
    opts = {k: HipAdamW(m.parameters(), lr=1e-5, grad_scale=m.loss_scale) for k, m in nets.items()}
    inv = {k: torch.full((), 0.5 * m.loss_scale, device=DEV) for k, m in nets.items()}
    for epoch in range(3):
        for name, m in nets.items():
            for acc in range(2):
                da, db = draws(100 + 10 * epoch + acc), draws(200 + 10 * epoch + acc)
                ta, tb = branch(m, da), branch(m, db)
                loss, dice = ops.consistency_loss(ta, tb, 1)
                hist[name].append((float(loss), dice.cpu()))
                torch.autograd.backward(loss, grad_tensors=inv[name])
            opts[name].step()
            opts[name].zero_grad()
    for (l32, d32), (l16, d16) in zip(hist["fp32"], hist["bf16"]):
        assert abs(l32 - l16) < 1e-3, f"consistency loss fp32 {l32:.5f} vs bf16 {l16:.5f}"
        assert (d32 - d16).abs().max() < 1e-3          # per-class soft Dice
    # final label maps and hard Dice against a pseudo ground truth (the fp32 prediction of the initial model)
    with torch.no_grad():
        noise = torch.randn(1, 12, 32, 32, 32, device=DEV)
        l32 = nets["fp32"](MIND3D()(imgs, noise))
        l16 = nets["bf16"](MIND3D()(imgs, noise, out_dtype=nets["bf16"].act_dtype))
    a32, a16 = l32.argmax(1), l16.argmax(1)
    assert (a32 == a16).float().mean() > (0.97 if low == "bf16" else 0.995)      # near-tied random-weight logits; bf16 measured 0.988
    gt = a32.roll(1, dims=-1)
    d32, d16 = dice_coeff(a32, gt, 8), dice_coeff(a16, gt, 8)
    # hard Dice of an UNTRAINED net is the worst case (near-tied logits: ~1.2 % of the voxels flip label under bf16
    # rounding); measured mean difference 1.1e-3, per class <= 3.6e-3.  The soft quantities above meet 1e-3.
    if low == "bf16":
        assert (d32 - d16).abs().max() < 6e-3 and abs(float(d32.nanmean()) - float(d16.nanmean())) < 2.5e-3
    else:       # fp16 storage meets north_star's 1e-3 on the hard Dice as well
        assert (d32 - d16).abs().max() < 1e-3 and abs(float(d32.nanmean()) - float(d16.nanmean())) < 1e-3


UNIT_MAPPING = {"background": (0, 0), "a": (2, 3), "b": (3, 1), "c": (5, 4), "d": (8, 2)}     # TTA ids != positions


@pytest.mark.parametrize("mode", ["batched", "sequential"])
@pytest.mark.parametrize("conv_impl", [0, 1])
def test_tta_unit_golden(mode, conv_impl, monkeypatch):
    """The PRODUCT loop `tta_unit` (default: 2 branches x 4 accumulation steps per network pass, in-place gradient
    accumulation, exact-zero bias gradients as tta_main configures it; and DGTTA_BATCH_BRANCHES=0) driven by the same
    CPU draw stream as the reference run that produced tests/golden/tta_unit.npz: 5 epochs x 8 steps at the plan's
    lr = 1e-5 (tta.py:189-340 with the reference's get_batch / calc_branch / soft_dice_loss / dice_coeff / AdamW).
    Per-epoch losses, pseudo-Dice, adapted parameters, and the final label map bit for bit."""
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.optim import HipAdamW
    from dg_tta_amd.tta.tta import _fuse_head_if_possible, tta_unit
    from dg_tta_amd.tta.torch_utils import get_batch, release_resident
    monkeypatch.setenv("DGTTA_BATCH_BRANCHES", "1" if mode == "batched" else "0")
    g = load_golden("tta_unit")
    model, modmod = _product_model(g, conv_impl=conv_impl)
    assert _fuse_head_if_possible(model, modmod, UNIT_MAPPING, OPTIMIZED)
    model.accumulate_grads_in_place = True
    model.exact_zero_bias_grad = True
    E, accum = int(g["epochs"]), int(g["accum"])
    cfg = _plan(epochs=E, patches_to_be_accumulated=accum, lr=float(g["lr"]))
    opt = HipAdamW(model.parameters(), lr=cfg["lr"])
    data = g["data"]
    release_resident()
    with cpu_rng_for_device_draws():
        torch.manual_seed(int(g["seed"]))
        np.random.seed(int(g["seed"]))
        losses, dices = tta_unit(model, opt, cfg, [data], [16, 16, 16], UNIT_MAPPING, modmod, torch.device(DEV), True)
    ref_l, ref_d = g["tta_losses"], g["eval_dices"]
    assert (losses - ref_l).abs().max() < 2e-5, f"epoch losses {losses.tolist()} vs reference {ref_l.tolist()}"
    # pseudo-Dice = hard Dice of the argmax against the mapped labels (counts): equal unless a near-tied voxel flips
    assert (dices - ref_d).abs().max() < 2e-4, f"pseudo-Dice {dices.tolist()} vs reference {ref_d.tolist()}"
    # adapted parameters: 4 AdamW steps of lr 1e-5; Adam's step is ~lr*sign(g) early on, so a parameter whose gradient
    # is rounding noise may move the other way (2 lr per step).  Everything else follows the reference.
    post, pre = state_from_golden(g, "p::"), state_from_golden(g, "w::")
    moved = agree = 0
    for name, p in model.state_dict().items():
        if name not in post or (name.endswith("conv.bias") and ".convs." in name):
            continue
        d_ref, d = post[name] - pre[name], p.cpu() - pre[name]
        assert (d - d_ref).abs().max() <= 8.5e-5, name          # never further than 2 lr per step apart
        moved += d_ref.numel()
        agree += int(((d - d_ref).abs() <= 0.1 * d_ref.abs() + 2e-7).sum())
    # measured 96.7 % (the rest: parameters whose accumulated gradient is at rounding-noise level, see the gradient
    # conditioning note in tests/test_gpu_full_topology.py)
    assert agree / moved > 0.95, f"only {agree / moved:.4f} of the parameter updates agree with the reference"
    # final label map of the adapted model: bit-exact (the golden draw has a minimal top-2 margin of 4e-4)
    model.set_selected_classes(get_map(UNIT_MAPPING))
    with torch.no_grad():
        imgs, _ = get_batch([data], [0], [16, 16, 16], "center", DEV)
        logits = model.forward(MIND3D()(imgs[0], g["eval_noise"].to(DEV)))      # .forward: no pre-hooks
    ref = g["eval_logits"]
    err = (logits.cpu() - ref).abs().max().item()
    # 4 AdamW steps: parameters whose gradient sign differs (see above) sit up to 8e-5 apart -> logits within ~1e-3
    assert err < 2e-3, f"final logits err {err:.3e} (min top-2 margin of the reference {g['eval_margin'].min():.3e})"
    assert torch.equal(logits.argmax(1).cpu(), g["eval_argmax"])
    # ... and on draws nobody picked (round 4): the FIRST noise draws of the generator's search (seeds 999..1002), scored by
    # the CPU oracle carrying the reference's post-TTA parameters.  Logits stay within the same bound; labels are identical
    # wherever the oracle's top-2 margin exceeds twice that bound, and overall agreement is reported against the mask.
    from oracle import mind as omind, tta as otta, unet as ounet
    omodel = ounet.PlainConvUNetOracle(SMALL_CFG)
    omodel.load_state_dict({**omodel.state_dict(), **post})
    omodel.eval()
    map_pre = otta.get_map_idxs(UNIT_MAPPING, OPTIMIZED, "pretrain_labels")
    cimgs = imgs[0].float().cpu()
    for nseed in (999, 1000, 1001, 1002):
        torch.manual_seed(nseed)
        noise = torch.randn(cimgs.shape[0], 12, 16, 16, 16)
        with torch.no_grad():
            oref = otta.map_label(omodel(omind.mind3d(cimgs, noise)), map_pre, "logits")
            got = model.forward(MIND3D()(imgs[0], noise.to(DEV))).cpu()
        assert (got - oref).abs().max().item() < 2e-3, nseed
        top2 = oref.topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 4e-3
        same = got.argmax(1) == oref.argmax(1)
        assert bool(same[safe].all()) and float(safe.float().mean()) > 0.9, (nseed, float(safe.float().mean()))
        assert float(same.float().mean()) > 0.995, (nseed, float(same.float().mean()))
    release_resident()


TRAINED_LIMITS = {
    # fixture: storage -> (epoch-loss tolerance, pseudo-Dice tolerance, |hard Dice vs GT - reference's|, min label agreement)
    # (limits = 2-3x the values measured on MI355X, profiles/r05_trained_unit_parity.json; the Dice tolerance is north_star's)
    # fp32 is held to north_star's clauses outright.  For 16-bit storage a 16^3 patch cannot resolve 1e-3: its structures hold
    # ~100 voxels, ONE flipped voxel moves a class Dice by 4e-3 and the mean by 1.1e-3 (fp16 flips 0-1 voxels of 4096, bf16
    # 4).  Their limits here are that granularity (<= 3 / <= 12 voxels); the 1e-3 claim for 16-bit storage is made where the
    # voxel count carries it: bench.py's dice_delta (128^3, against the CPU oracle's run) and tests/test_gpu_at_size.py.
    "tta_unit_trained": {None: (2e-5, 1e-3, 1e-3, 1.0), torch.float16: (6e-3, 8e-3, 3.3e-3, 0.9992), torch.bfloat16: (1.5e-2, 2.5e-2, 1.3e-2, 0.997)},
    "tta_unit_trained_mind": {None: (2e-5, 1e-3, 1e-3, 1.0), torch.float16: (6e-3, 8e-3, 3.3e-3, 0.9992), torch.bfloat16: (1.5e-2, 2.5e-2, 1.3e-2, 0.997)},
}


@pytest.mark.parametrize("storage", [None, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("fixture", ["tta_unit_trained", "tta_unit_trained_mind"])
def test_tta_unit_trained_golden(fixture, storage):
    """Round 5: the product loop against a REFERENCE run that means something - a net PRE-TRAINED on the
    source domain of the synthetic atlas task (hard Dice 0.81 / 0.88 on unseen source cases), adapted for 12 epochs x 8 steps at
    lr 3e-4 to a case of the shifted target domain by the reference's own loop (tests/golden/make_golden_r5.py: tta.py:189-340
    around its real get_batch / calc_branch / soft_dice_loss / dice_coeff, torch AdamW).  `tta_unit_trained`: GIN + MIND
    pre-training, Dice vs ground truth 0.821 -> 0.807; `tta_unit_trained_mind`: MIND-only pre-training on a low-SNR target,
    0.693 -> 0.713 (adaptation wins back a tenth of the gap).  Checked: per-epoch consistency loss, pseudo-Dice of the
    evaluation patch, HARD DICE VS GROUND TRUTH after adaptation within north_star's 1e-3 of the reference's for every storage
    type, and - fp32 - the final label map BIT FOR BIT on the first MIND noise draw after the run (no search for a convenient
    draw; the reference's minimal top-2 margin on it is 2e-2 / 2e-3)."""
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.optim import HipAdamW
    from dg_tta_amd.tta.tta import _fuse_head_if_possible, tta_unit
    from dg_tta_amd.tta.torch_utils import dice_coeff, get_batch, get_map_idxs, map_label, release_resident
    g = load_golden(fixture)
    kw = {} if storage is None else {"act_dtype": storage}
    model, modmod = _product_model(g, conv_impl=0, **kw)
    assert _fuse_head_if_possible(model, modmod, UNIT_MAPPING, OPTIMIZED)
    model.accumulate_grads_in_place = True
    model.exact_zero_bias_grad = True
    E, accum = int(g["epochs"]), int(g["accum"])
    cfg = _plan(epochs=E, patches_to_be_accumulated=accum, lr=float(g["lr"]))
    opt = HipAdamW(model.parameters(), lr=cfg["lr"], grad_scale=model.loss_scale)
    data = g["data"]
    release_resident()
    with cpu_rng_for_device_draws():
        torch.manual_seed(int(g["seed"]))
        np.random.seed(int(g["seed"]))
        losses, dices = tta_unit(model, opt, cfg, [data], [16, 16, 16], UNIT_MAPPING, modmod, torch.device(DEV), True)
    assert int(opt.skipped_steps) == 0
    tol_l, tol_pd, tol_d, min_agree = TRAINED_LIMITS[fixture][storage]
    ref_l, ref_d = g["tta_losses"], g["eval_dices"]
    dl, dd = float((losses - ref_l).abs().max()), float((dices - ref_d).abs().max())
    # final prediction on the stored first noise draw, Dice against the case's own labels
    model.set_selected_classes(get_map(UNIT_MAPPING))
    with torch.no_grad():
        imgs, labels = get_batch([data], [0], [16, 16, 16], "center", DEV)
        logits = model.forward(MIND3D()(imgs[0], g["eval_noise"].to(DEV), **({} if storage is None else {"out_dtype": storage})))
        gt = map_label(labels[0], get_map_idxs(UNIT_MAPPING, OPTIMIZED, "tta_labels"), "argmaxed").long()
        per_class = dice_coeff(logits.argmax(1), gt, len(OPTIMIZED))
    ref_after, ref_before = g["dice_after"], g["dice_before"]
    agree = float((logits.argmax(1).cpu() == g["eval_argmax"]).float().mean())
    err = float((logits.float().cpu() - g["eval_logits"]).abs().max())
    d_mean = abs(float(per_class.nanmean()) - float(ref_after.nanmean()))
    d_cls = float((per_class.cpu() - ref_after).abs().max())
    print(f"\n{fixture} {storage}: loss delta {dl:.3e}, pseudo-Dice delta {dd:.3e}, hard Dice vs GT {float(per_class.nanmean()):.4f} "
          f"(reference {float(ref_after.nanmean()):.4f}, before TTA {float(ref_before.nanmean()):.4f}): mean delta {d_mean:.2e}, "
          f"per-class max {d_cls:.2e}; labels equal {agree:.6f}, logit err {err:.2e}")
    assert dl < tol_l, f"epoch losses {losses.tolist()} vs reference {ref_l.tolist()}"
    assert dd < tol_pd, f"pseudo-Dice {dices.tolist()} vs reference {ref_d.tolist()}"
    assert d_mean <= tol_d, f"hard Dice vs GT {per_class.tolist()} vs the reference's {ref_after.tolist()}"
    assert float(ref_after.nanmean()) >= 0.5 and abs(float(ref_after.nanmean()) - float(ref_before.nanmean())) > 1e-2
    assert agree >= min_agree
    if storage is None:
        assert torch.equal(logits.argmax(1).cpu(), g["eval_argmax"])        # bit for bit, first draw
    release_resident()


def get_map(mapping):
    from dg_tta_amd.tta.torch_utils import get_map_idxs
    return get_map_idxs(mapping, OPTIMIZED, "pretrain_labels")


def test_evaluation_targets_use_the_optimized_label_index_space(tmp_path):
    """tta.py:440-447: mapped_target_labels* hold the TTA dataset's ids mapped to positions in optimized_labels (unmapped
    ids -> 0), so summary_*.json compares like with like when the TTA ids are not 0..N-1 in list order."""
    import json
    from types import SimpleNamespace as NS
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.tta import tta_main
    g = load_golden("calc_branch")
    net = _network_with_hooks(g, conv_impl=0)
    modmod = SimpleNamespace(ModifierFunctions=ModifierFunctions)
    mapping = {"background": (0, 0), "a": (2, 3), "b": (3, 1)}           # TTA label 3 is 'a' (position 1), 1 is 'b'
    cfg = _plan(epochs=1, ensemble_count=1, patches_to_be_accumulated=2, tta_data_filepaths=[], seed=3,
                pretrained_weights_filepath="unused", optimized_labels=["background", "a", "b"])
    case = _synthetic_case(5)                                            # 3 one-hot label channels = TTA ids 1, 2, 3
    raw = torch.cat([(case[1:].sum(0, keepdim=True) < 1).float(), case[1:]]).argmax(0)
    bundle = (NS(), [16, 16, 16], net, [{k: v.clone() for k, v in net.state_dict().items()}])
    data = (iter([{"data": case, "data_properties": {}, "ofile": "tta_outputTs/case5"}]), 1)
    tta_main("run0", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data)
    tgt = torch.from_numpy(np.load(tmp_path / "run0" / "mapped_target_labelsTs" / "case5.npy").astype(np.int64))
    expect = torch.zeros_like(raw)
    expect[raw == 3] = 1        # 'a'
    expect[raw == 1] = 2        # 'b'; TTA id 2 is not optimised -> background
    assert torch.equal(tgt, expect)
    sj = json.loads((tmp_path / "run0" / "summary_Ts.json").read_text())
    assert sj["metric_per_case"][0]["metrics"]["1"]["n_ref"] == int((raw == 3).sum())
    assert sj["metric_per_case"][0]["metrics"]["2"]["n_ref"] == int((raw == 1).sum())


def test_two_ranks_one_run_directory(tmp_path):
    """Multi-GPU correctness without a second GPU: ranks are run one after the other on the same run directory.
    (a) 2 samples x 2 members on 2 ranks: rank 0 may only write the summary once rank 1's marker exists, and the summary
    then contains every case; (b) 1 sample x 2 members on 2 ranks: the members are adapted on different ranks, the owner
    of member 0 waits for the other member's parameter file and predicts with both."""
    import json
    from types import SimpleNamespace as NS
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.tta import tta_main
    g = load_golden("calc_branch")
    net = _network_with_hooks(g, conv_impl=0)
    modmod = SimpleNamespace(ModifierFunctions=ModifierFunctions)
    mapping = {"background": (0, 0), "a": (2, 1), "b": (3, 2), "c": (5, 3)}
    cfg = _plan(epochs=2, ensemble_count=2, patches_to_be_accumulated=2, tta_data_filepaths=[], seed=3,
                pretrained_weights_filepath="unused", lr=1e-4, optimized_labels=["background", "a", "b", "c"],
                barrier_timeout_s=1.0)
    bundle = (NS(), [16, 16, 16], net, [{k: v.clone() for k, v in net.state_dict().items()}])

    def data(seeds):
        return iter([{"data": _synthetic_case(s), "data_properties": {}, "ofile": f"tta_outputTs/case{s}"}
                     for s in seeds]), len(seeds)

    # (a) rank 0 alone: the barrier must time out instead of summarising half a run
    with pytest.raises(TimeoutError):
        tta_main("runA", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data((1, 2)),
                 shard=(0, 2))
    assert not (tmp_path / "runA" / "summary_Ts.json").exists()
    r1 = tta_main("runA", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data((1, 2)),
                  shard=(1, 2))
    assert not any(k[0] == "summary" for k in r1)
    r0 = tta_main("runA", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data((1, 2)),
                  shard=(0, 2))
    assert ("summary", "Ts") in r0
    sj = json.loads((tmp_path / "runA" / "summary_Ts.json").read_text())
    assert sorted(Path(c["prediction_file"]).name for c in sj["metric_per_case"]) == ["case1.npy", "case2.npy"]
    # (b) fewer samples than ranks: members spread over the ranks
    with pytest.raises(TimeoutError):       # member 1 lives on rank 1: rank 0 cannot predict yet
        tta_main("runB", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data((4,)),
                 shard=(0, 2))
    out = tmp_path / "runB" / "tta_outputTs"
    assert (out / "case4__ensemble_idx_0_tta_parameters.pt").is_file()
    assert not (out / "case4__ensemble_idx_1_tta_parameters.pt").exists()
    tta_main("runB", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data((4,)), shard=(1, 2))
    assert (out / "case4__ensemble_idx_1_tta_parameters.pt").is_file() and not (tmp_path / "runB" / "tta_outputTs" / "case4.npy").exists()
    rb = tta_main("runB", cfg, tmp_path, tmp_path, mapping, modmod, DEV, network_bundle=bundle, tta_data=data((4,)),
                  shard=(0, 2))
    assert ("tta_outputTs/case4", "prediction") in rb and ("summary", "Ts") in rb


def test_side_streams_do_not_change_a_bit(monkeypatch):
    """The weight-gradient side stream (unet backward) and the input pipeline of tta_epoch (next pass prepared on a third
    stream, MIND precomputed) against the one-stream schedule: same seeds -> identical losses, pseudo-Dice and adapted
    parameters, bit for bit (draws happen at enqueue time in the same order; accumulation orders are fixed per stream)."""
    from dg_tta_amd.optim import HipAdamW
    from dg_tta_amd.tta.tta import _fuse_head_if_possible, tta_unit
    from dg_tta_amd.tta.torch_utils import release_resident
    g = load_golden("tta_unit")

    def run(streams):
        monkeypatch.setenv("DGTTA_WGRAD_STREAM", streams)
        monkeypatch.setenv("DGTTA_PIPELINE_PREP", streams)
        model, modmod = _product_model(g, conv_impl=0, act_dtype=torch.bfloat16)
        assert _fuse_head_if_possible(model, modmod, UNIT_MAPPING, OPTIMIZED)
        model.accumulate_grads_in_place = True
        model.exact_zero_bias_grad = True
        cfg = _plan(epochs=4, patches_to_be_accumulated=8, lr=1e-4)
        opt = HipAdamW(model.parameters(), lr=cfg["lr"])
        release_resident()
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        np.random.seed(77)
        losses, dices = tta_unit(model, opt, cfg, [g["data"]], [16, 16, 16], UNIT_MAPPING, modmod, torch.device(DEV), True)
        torch.cuda.synchronize()
        release_resident()
        return losses, dices, {k: v.clone() for k, v in model.state_dict().items()}

    l1, d1, p1 = run("1")
    l0, d0, p0 = run("0")
    assert torch.equal(l1, l0) and torch.equal(d1, d0)
    assert all(torch.equal(p1[k], p0[k]) for k in p1)
    assert float((l1[1:] - l1[:-1]).abs().max()) > 0          # the runs did adapt


def test_two_instances_on_two_threads_match_their_sequential_runs():
    """Per-model host state: two tta_units in ONE process, each on its own thread, stream and draw
    generators (utils.rng_scope), give bit-identical losses, Dice and parameters to the same two units run one after the
    other - nothing (MIND hand-over, side streams, draw order) leaks from one instance into the other."""
    import threading
    from dg_tta_amd.optim import HipAdamW
    from dg_tta_amd.tta.tta import _fuse_head_if_possible, tta_unit
    from dg_tta_amd.utils import rng_scope
    g = load_golden("tta_unit")

    def unit(seed, stream, out):
        try:
            with torch.cuda.stream(stream):
                model, modmod = _product_model(g, conv_impl=0, act_dtype=torch.float16)
                assert _fuse_head_if_possible(model, modmod, UNIT_MAPPING, OPTIMIZED)
                model.accumulate_grads_in_place = True
                model.exact_zero_bias_grad = True
                cfg = _plan(epochs=3, patches_to_be_accumulated=8, lr=1e-4)
                opt = HipAdamW(model.parameters(), lr=cfg["lr"], grad_scale=model.loss_scale)
                data = [g["data"].clone() + 0.01 * seed]          # an instance's own case
                gens = (torch.Generator().manual_seed(seed), torch.Generator(device=DEV).manual_seed(seed),
                        np.random.RandomState(seed))
                with rng_scope(*gens):
                    losses, dices = tta_unit(model, opt, cfg, data, [16, 16, 16], UNIT_MAPPING, modmod, torch.device(DEV), True)
                stream.synchronize()
                out[seed] = (losses, dices, {k: v.detach().clone() for k, v in model.state_dict().items()})
        except BaseException as e:        # surfaced in the main thread
            out[seed] = e

    seq, par = {}, {}
    for seed in (5, 6):
        unit(seed, torch.cuda.Stream(), seq)
    threads = [threading.Thread(target=unit, args=(seed, torch.cuda.Stream(), par)) for seed in (5, 6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for seed in (5, 6):
        for res in (seq[seed], par[seed]):
            if isinstance(res, BaseException):
                raise res
        (l0, d0, p0), (l1, d1, p1) = seq[seed], par[seed]
        assert torch.equal(l0, l1) and torch.equal(d0, d1), f"instance {seed}: threaded run differs"
        assert all(torch.equal(p0[k], p1[k]) for k in p0)
    assert not torch.equal(seq[5][0], seq[6][0])                   # the two instances are different problems


def test_mind_is_not_precomputed_behind_a_user_input_modifier():
    """tta_epoch evaluates MIND ahead of the network call only when nothing in front of mind_hook can change the input: a
    user-defined modify_tta_input_fn (or the trainers' internal GIN augmentation) switches the shortcut off."""
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.model_utils import get_model_from_network
    from dg_tta_amd.tta.tta import _can_precompute_mind
    from dg_tta_amd import utils
    g = load_golden("tta_unit")
    model, modmod = _product_model(g)
    assert _can_precompute_mind(model, modmod)

    class Custom(ModifierFunctions):
        @staticmethod
        def modify_tta_input_fn(image: torch.Tensor):
            return image * 2.0

    custom = SimpleNamespace(ModifierFunctions=Custom)
    model2 = get_model_from_network(_network_with_hooks(g), custom, None)
    assert not _can_precompute_mind(model2, custom)
    import os
    old = os.environ.get("DG_TTA_INTERNAL_AUGMENTATION")
    try:
        os.environ["DG_TTA_INTERNAL_AUGMENTATION"] = "true"
        assert not _can_precompute_mind(model, modmod)
    finally:
        if old is None:
            os.environ.pop("DG_TTA_INTERNAL_AUGMENTATION", None)
        else:
            os.environ["DG_TTA_INTERNAL_AUGMENTATION"] = old
    utils.disable_internal_augmentation()


def test_each_rank_loads_only_the_cases_it_works_on(tmp_path, monkeypatch):
    """Sharded tta_main through the REAL data iterator: 3 cases on 2 ranks - a rank reads and
    preprocesses only its own cases (the others arrive as stubs), predicts each case right after its last ensemble member
    and drops it; the summary written after both ranks covers all three."""
    import json
    from types import SimpleNamespace as NS
    from dg_tta_amd.tta import nnunet_utils as nu
    from dg_tta_amd.tta.config_log_utils import ModifierFunctions
    from dg_tta_amd.tta.tta import tta_main
    raw = tmp_path / "raw"
    (raw / "imagesTs").mkdir(parents=True)
    (raw / "labelsTs").mkdir()
    files = []
    for s in (1, 2, 3):
        case = _synthetic_case(s)
        np.save(raw / "imagesTs" / f"c{s}_0000.npy", case[:1].numpy())
        lab = torch.cat([(case[1:].sum(0, keepdim=True) < 1).float(), case[1:]]).argmax(0)
        np.save(raw / "labelsTs" / f"c{s}.npy", lab.numpy().astype(np.int16))
        files.append(str(raw / "imagesTs" / f"c{s}_0000.npy"))
    g = load_golden("calc_branch")
    net = _network_with_hooks(g, conv_impl=0)
    modmod = SimpleNamespace(ModifierFunctions=ModifierFunctions)
    mapping = {"background": (0, 0), "a": (2, 1), "b": (3, 2), "c": (5, 3)}
    cfg = _plan(epochs=1, start_tta_at_epoch=0, ensemble_count=2, patches_to_be_accumulated=2, tta_data_filepaths=files, seed=3,
                pretrained_weights_filepath="unused", lr=1e-4, optimized_labels=["background", "a", "b", "c"],
                barrier_timeout_s=60.0)
    bundle = (NS(), [16, 16, 16], net, [{k: v.clone() for k, v in net.state_dict().items()}])
    read = []
    real = nu.preprocess_fromfile
    monkeypatch.setattr(nu, "preprocess_fromfile", lambda f, *a, **k: (read.append(Path(f).name), real(f, *a, **k))[1])
    out = tmp_path / "out"
    out.mkdir()
    r1 = tta_main("run", cfg, raw, out, mapping, modmod, DEV, network_bundle=bundle, shard=(1, 2))
    assert read == ["c2_0000.npy"]                                        # rank 1 of 2: sample index 1 only
    assert ("tta_outputTs/c2", "prediction") in r1 and not any(k[0] == "summary" for k in r1)
    read.clear()
    r0 = tta_main("run", cfg, raw, out, mapping, modmod, DEV, network_bundle=bundle, shard=(0, 2))
    assert read == ["c1_0000.npy", "c3_0000.npy"]                         # rank 0: indices 0 and 2, each read once
    assert ("summary", "Ts") in r0
    sj = json.loads((out / "run" / "summary_Ts.json").read_text())
    assert sorted(Path(c["prediction_file"]).name for c in sj["metric_per_case"]) == ["c1.npy", "c2.npy", "c3.npy"]
