"""CPU: the oracle reproduces every golden vector generated from the reference (tests/golden/make_golden.py)."""
import pytest
import torch

from conftest import load_golden, unpack_draws, SMALL_CFG, state_from_golden
from oracle import mind as omind, gin as ogin, tta as otta, unet as ounet


def test_mind_golden():
    for tag in ("16", "ragged", "const"):
        g = load_golden(f"mind3d_{tag}")
        assert torch.equal(omind.mind3d(g["img"], g["noise"]), g["out"])


def test_mind_constants():
    taps = omind.gauss_taps(1.0)
    ref = torch.tensor([0.054488689, 0.244201362, 0.402619958, 0.244201362, 0.054488689])
    assert torch.allclose(taps, ref, atol=1e-8)
    assert len(omind.SHIFT1) == len(omind.SHIFT2) == 12


def test_gin_golden():
    for i in range(7):
        g = load_golden(f"gin_{i}")
        ks = [int(k) for k in g["ks"]]
        out = ogin.gin_chain(g["x"], g["alpha"], ks, [g[f"ker{j}"] for j in range(4)],
                             [g[f"shift{j}"] for j in range(4)])
        assert torch.equal(out, g["out"])


def test_loss_golden():
    g = load_golden("loss")
    assert torch.equal(otta.soft_dice_loss(g["a"], g["b"]), g["d_ab"])
    assert torch.equal(otta.soft_dice_loss(g["a"], g["a"]), g["d_aa"])
    assert torch.equal(otta.consistency_loss(g["ta"], g["tb"]), torch.as_tensor(g["loss"]))
    assert torch.equal(otta.dice_coeff(g["dc_out"], g["dc_lab"], 4), g["dc"])
    z = torch.zeros(1, 3, 4, 4, 4)
    assert torch.equal(otta.soft_dice_loss(z, z), torch.ones(1, 3))


def test_mapping_golden():
    g = load_golden("mapping")
    assert torch.equal(otta.map_label(g["logits"], g["idx"], "logits"), g["mapped"])
    assert torch.equal(otta.map_label(g["am"], g["idx"], "argmaxed"), g["am_mapped"])


def test_get_batch_golden():
    g = load_golden("get_batch")
    img, lbl = otta.get_batch_item(g["data"], [16, 16, 16], g["rand3"])
    assert torch.equal(img, g["img"]) and torch.equal(lbl, g["lbl"])
    img, lbl = otta.get_batch_item(g["data"], [16, 16, 16], None)
    assert torch.equal(img, g["cimg"]) and torch.equal(lbl, g["clbl"])
    # centre crop of an even margin is a plain crop (up to the (x-min)+min rounding of torch_utils.py:58-62)
    assert torch.allclose(img[0, 0], g["data"][0, 2:18, 1:17, 3:19], rtol=0, atol=2e-3)
    img, lbl = otta.get_batch_item(g["small"], [16, 16, 16], g["small_rand3"])
    assert lbl is None and torch.equal(img, g["small_img"])


def test_rand_affine_golden():
    g = load_golden("rand_affine")
    r, rinv = otta.rand_affine_from_draw(g["draw"])
    assert torch.equal(r, g["r"]) and torch.equal(rinv, g["rinv"])


def test_calc_branch_golden():
    g = load_golden("calc_branch")
    m = ounet.PlainConvUNetOracle(SMALL_CFG)
    m.load_state_dict(state_from_golden(g), strict=False)
    for br in ("a", "b"):
        out = otta.calc_branch(m, g["imgs"], g["map_idxs"], **unpack_draws(g, br))
        assert torch.allclose(out, g[f"out_{br}"], rtol=0, atol=2e-5)  # thread-count dependent conv summation


def test_tta_epoch_golden():
    g = load_golden("tta_epoch")
    m = ounet.PlainConvUNetOracle(SMALL_CFG)
    m.load_state_dict(state_from_golden(g), strict=False)
    opt = torch.optim.AdamW(m.parameters(), lr=float(g["lr"]))
    losses = []
    for p in m.parameters():
        p.requires_grad_(False)
    for epoch in range(3):
        if epoch == 1:
            for p in m.parameters():
                p.requires_grad_(True)
        for acc in range(2):
            losses.append(otta.tta_step(m, g["imgs"], g["map_idxs"], unpack_draws(g, f"e{epoch}s{acc}_a"),
                                        unpack_draws(g, f"e{epoch}s{acc}_b"), accum=2, backward=epoch >= 1))
        if epoch >= 1:
            opt.step(), opt.zero_grad()
    assert torch.allclose(torch.stack(losses), g["losses"], rtol=0, atol=1e-5)
    with torch.no_grad():
        logits = otta.map_label(m(omind.mind3d(g["imgs"], g["eval_noise"])), g["map_idxs"], "logits")
    assert (logits.argmax(1) == g["eval_argmax"]).float().mean() > 0.999


def test_unet_keys_and_size():
    m = ounet.PlainConvUNetOracle()
    assert abs(sum(p.numel() for p in m.parameters()) - 16_606_948) == 0
    keys = m.state_dict().keys()
    assert "encoder.stages.0.0.convs.0.all_modules.1.weight" in keys
    assert "decoder.encoder.stages.4.0.convs.1.conv.bias" in keys
    assert "decoder.transpconvs.3.weight" in keys and "decoder.seg_layers.0.bias" in keys


def test_tta_unit_golden_cpu():
    """The oracle's restatement of the WHOLE unit loop (tta.py:189-340) reproduces the reference run bit for bit: losses,
    pseudo-Dice per epoch, post-TTA parameters and the final label map (plan lr 1e-5, 5 epochs x 8 steps)."""
    import numpy as np
    g = load_golden("tta_unit")
    om = ounet.PlainConvUNetOracle(SMALL_CFG)
    om.load_state_dict(state_from_golden(g, "w::"), strict=False)
    lm = {"background": (0, 0), "a": (2, 3), "b": (3, 1), "c": (5, 4), "d": (8, 2)}
    names = ["background", "a", "b", "c", "d"]
    pre, tta = otta.get_map_idxs(lm, names, "pretrain_labels"), otta.get_map_idxs(lm, names, "tta_labels")
    opt = torch.optim.AdamW(om.parameters(), lr=float(g["lr"]))
    torch.manual_seed(int(g["seed"]))
    np.random.seed(int(g["seed"]))
    losses, dices, steps = otta.tta_unit(om, opt, [g["data"]], [16, 16, 16], pre, tta, int(g["epochs"]), 1, int(g["accum"]))
    # bit-identical with the generating thread count (make_golden_r2.py asserts torch.equal); the summation order of
    # torch's CPU kernels depends on the thread count, hence a float tolerance here
    assert torch.allclose(steps, g["step_losses"], atol=2e-6) and torch.allclose(losses, g["tta_losses"], atol=2e-6)
    assert torch.allclose(dices, g["eval_dices"], atol=1e-6)
    post = state_from_golden(g, "p::")
    for k, v in om.state_dict().items():
        if k in post and not (k.endswith("conv.bias") and ".convs." in k):     # zero-gradient biases: Adam on noise
            assert torch.allclose(v, post[k], atol=2.5e-5), k
    with torch.no_grad():
        imgs, _ = otta.get_batch_item(g["data"], [16, 16, 16], None)
        final = otta.map_label(om(omind.mind3d(imgs, g["eval_noise"])), pre, "logits")
    assert torch.equal(final.argmax(1), g["eval_argmax"])


@pytest.mark.parametrize("fixture", ["tta_unit_trained", "tta_unit_trained_mind"])
def test_tta_unit_trained_golden_cpu(fixture):
    """Round 5: the oracle reproduces the reference's run on PRE-TRAINED weights under a domain shift (make_golden_r5.py: 12
    epochs x 8 steps at lr 3e-4; consistency loss, pseudo-Dice, adapted parameters) and its before / after hard Dice vs
    ground truth on the stored first noise draw; the fixture itself says the run means something (source Dice >= 0.8, target
    Dice >= 0.5, adaptation moves it by more than ten times north_star's tolerance)."""
    import numpy as np
    g = load_golden(fixture)
    om = ounet.PlainConvUNetOracle(SMALL_CFG)
    om.load_state_dict(state_from_golden(g, "w::"), strict=False)
    lm = {"background": (0, 0), "a": (2, 3), "b": (3, 1), "c": (5, 4), "d": (8, 2)}
    names = ["background", "a", "b", "c", "d"]
    pre, tta = otta.get_map_idxs(lm, names, "pretrain_labels"), otta.get_map_idxs(lm, names, "tta_labels")

    def hard_dice(model):
        with torch.no_grad():
            imgs, labels = otta.get_batch_item(g["data"], [16, 16, 16], None)
            out = otta.map_label(model(omind.mind3d(imgs, g["eval_noise"])), pre, "logits")
            return otta.dice_coeff(out.argmax(1), otta.map_label(labels, tta, "argmaxed").long(), len(names)), out
    before, _ = hard_dice(om)
    assert torch.allclose(before, g["dice_before"], atol=1e-6)
    assert float(g["source_dice"].nanmean()) >= 0.8 and float(g["dice_before"].nanmean()) >= 0.5
    assert abs(float(g["dice_after"].nanmean()) - float(g["dice_before"].nanmean())) > 1e-2
    opt = torch.optim.AdamW(om.parameters(), lr=float(g["lr"]))
    torch.manual_seed(int(g["seed"]))
    np.random.seed(int(g["seed"]))
    losses, dices, steps = otta.tta_unit(om, opt, [g["data"]], [16, 16, 16], pre, tta, int(g["epochs"]), 1, int(g["accum"]))
    # (thread-count dependent summation order of torch's CPU kernels: float tolerance, see test_tta_unit_golden_cpu)
    assert torch.allclose(steps, g["step_losses"], atol=2e-5) and torch.allclose(losses, g["tta_losses"], atol=2e-5)
    assert torch.allclose(dices, g["eval_dices"], atol=1e-3)
    om.eval()
    after, final = hard_dice(om)
    assert float((after - g["dice_after"]).abs().max()) <= 2e-3
    assert float((final.argmax(1) == g["eval_argmax"]).float().mean()) >= 0.9995


def test_full_topology_golden_cpu():
    """Full nnUNet 3d_fullres (32..320 features, 12 -> 105 channels) at 32^3: the oracle reproduces the reference's
    calc_branch x 2 + loss + backward (strided logit slices, label map, loss, gradient checksums)."""
    g = load_golden("full_32")
    torch.set_num_threads(8)
    om = ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(), int(g["w_seed"])), int(g["w_seed"]) + 1)
    sel = torch.arange(int(g["copt"])) * 3
    torch.manual_seed(int(g["img_seed"]))
    imgs = torch.randn(1, 1, 32, 32, 32)
    assert abs(imgs.double().sum().item() - float(g["imgs_sum"])) < 1e-9
    outs = {}
    for br in ("a", "b"):
        torch.manual_seed(int(g[f"seed_{br}"]))
        outs[br] = otta.calc_branch(om, imgs, sel, **otta.draw_branch(1, [32, 32, 32]))
        st = int(g["slice_step"])
        ref = g[f"out_{br}_slice"]
        assert (outs[br][:, :, ::st, ::st, ::st] - ref).abs().max() < 2e-5 * ref.abs().max()
        safe = g[f"out_{br}_margin"].float() > 1e-3
        assert torch.equal(outs[br].argmax(1)[safe].to(torch.uint8), g[f"out_{br}_argmax"][safe])
    loss = otta.consistency_loss(outs["a"], outs["b"])
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    loss.backward()
    for name, p in om.named_parameters():
        if f"g::{name}" in g and name.endswith("norm.weight"):
            ref = g[f"g::{name}"]
            assert (p.grad - ref).abs().max() < 2e-4 * ref.abs().max() + 1e-9, name
