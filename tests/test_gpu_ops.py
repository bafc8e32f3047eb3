"""GPU parity: HIP kernels (through the C ABI) vs the CPU oracle and the reference-generated golden vectors.

Tolerances (fp32 path): the kernels and the oracle sum in different orders, so float results agree to a few ulp of
the accumulated magnitude; stated per test.  Integer outputs (argmax, label patches) must be bit-exact.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, SMALL_CFG, state_from_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _close(a, b, atol, rtol=0.0, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    lim = atol + rtol * b.abs()
    assert bool((err <= lim).all()), f"{what}: max err {err.max().item():.3e} (limit {atol:g}+{rtol:g}*|ref|), " \
                                     f"ref max {b.abs().max().item():.3e}"


# ------------------------------------------------------------------------------------------------ MIND
@pytest.mark.parametrize("tag", ["16", "ragged", "const"])
def test_mind_golden(tag):
    from dg_tta_amd import ops
    g = load_golden(f"mind3d_{tag}")
    out = ops.mind3d(g["img"].to(DEV), g["noise"].to(DEV))
    _close(out, g["out"], atol=2e-6, rtol=2e-5, what=f"mind3d {tag}")     # outputs are exp(-x) in (0,1]


def test_mind_multitile_and_layouts():
    from dg_tta_amd import ops
    from oracle import mind as omind
    torch.manual_seed(3)
    img = torch.randn(1, 1, 19, 21, 70) * 3 + 1      # several tiles per axis, ragged edges
    noise = torch.randn(1, 12, 19, 21, 70)
    ref = omind.mind3d(img, noise)
    out = ops.mind3d(img.to(DEV), noise.to(DEV))
    _close(out, ref, atol=2e-6, rtol=2e-5, what="mind3d multitile")
    nd = ops.mind3d(img.to(DEV), noise.to(DEV), out_format="ndhwc", out_ldc=16)
    assert tuple(nd.shape) == (1, 19, 21, 70, 16)
    assert torch.equal(nd[..., :12].permute(0, 4, 1, 2, 3).contiguous(), out)       # same numbers, other layout
    assert float(nd[..., 12:].abs().max()) == 0.0
    bf = ops.mind3d(img.to(DEV), noise.to(DEV), out_format="ndhwc", out_ldc=16, out_dtype=torch.bfloat16)
    _close(bf[..., :12].float(), nd[..., :12], atol=4e-3, what="mind3d bf16 store")


def test_mind_module_draws_device_noise():
    from dg_tta_amd.mind import MIND3D, mind_hook
    x = torch.randn(1, 1, 16, 16, 16, device=DEV)
    torch.manual_seed(5)
    a = MIND3D()(x)
    torch.manual_seed(5)
    n = torch.randn(1, 12, 16, 16, 16, device=DEV)
    b = MIND3D()(x, n)
    assert torch.equal(a, b)
    torch.manual_seed(5)
    assert torch.equal(mind_hook(None, (x,)), a)


@pytest.mark.parametrize("delta,sigma", [(1, 1.0), (2, 1.0), (1, 0.6), (1, 2.0), (2, 1.7)])
def test_mind_delta_sigma_variants(delta, sigma):
    """MIND3D(delta, sigma) other than the reference's default (mind.py:98-140): neighbour distance 1 / 2 (replicate
    padding by delta, dilated shifts), Gaussian of 3 / 5 / 7 taps, on a ragged multi-tile volume against the oracle."""
    from dg_tta_amd.mind import MIND3D
    from oracle import mind as omind
    torch.manual_seed(int(10 * sigma) + delta)
    img = torch.randn(2, 1, 19, 23, 41) * 1.5
    noise = torch.randn(2, 12, 19, 23, 41)
    ref = omind.mind3d(img, noise, delta=delta, sigma=sigma)
    out = MIND3D(delta=delta, sigma=sigma)(img.to(DEV), noise.to(DEV))
    _close(out, ref, atol=3e-5, what=f"mind3d delta={delta} sigma={sigma}")
    with pytest.raises(NotImplementedError):
        MIND3D(delta=3)
    with pytest.raises(NotImplementedError):
        MIND3D(sigma=2.5)


# ------------------------------------------------------------------------------------------------ GIN
@pytest.mark.parametrize("i", range(7))
def test_gin_golden(i):
    from dg_tta_amd import ops
    g = load_golden(f"gin_{i}")
    ks = [int(k) for k in g["ks"]]
    out = ops.gin_chain(g["x"].to(DEV), g["alpha"].to(DEV), ks, [g[f"ker{j}"].to(DEV) for j in range(4)],
                        [g[f"shift{j}"].to(DEV) for j in range(4)])
    scale = float(g["out"].abs().max())
    _close(out, g["out"], atol=2e-5 * scale, what=f"gin {i} ks={ks}")


def test_gin_multitile():
    from dg_tta_amd import ops
    from oracle import gin as ogin
    torch.manual_seed(9)
    x = torch.randn(2, 1, 21, 19, 37)
    torch.manual_seed(10)
    alpha, ks, kers, shifts = ogin.draw_gin_params(2)
    ref = ogin.gin_chain(x, alpha, ks, kers, shifts)
    out = ops.gin_chain(x.to(DEV), alpha.to(DEV), ks, [k.to(DEV) for k in kers], [s.to(DEV) for s in shifts])
    _close(out, ref, atol=2e-5 * float(ref.abs().max()), what="gin multitile")
    # Frobenius norm preservation (gin.py:197-228): ||out|| == ||x|| per sample
    for b in range(2):
        assert abs(float(out[b].norm()) / float(x[b].norm()) - 1.0) < 1e-4


def test_gin_aug_draw_order_matches_reference():
    from dg_tta_amd.gin import draw_gin_params
    from oracle import gin as ogin
    torch.manual_seed(77)
    a1, k1, w1, s1 = ogin.draw_gin_params(1)
    torch.manual_seed(77)
    torch.rand(1)                       # the oracle drew alpha from the CPU generator; the device draw does not touch it
    a2, k2, w2, s2 = draw_gin_params(1, DEV)
    assert k1 == k2
    for p, q in zip(w1 + s1, w2 + s2):
        assert torch.equal(p, q.cpu())


# ------------------------------------------------------------------------------------------------ warp / sampling
def _theta(b, seed, strength=0.05):
    from oracle import tta as otta
    torch.manual_seed(seed)
    return otta.rand_affine_from_draw(torch.randn(b, 3, 4), strength)


@pytest.mark.parametrize("pad", ["border", "zeros"])
@pytest.mark.parametrize("cl,ch", [(False, 4), (True, 4), (True, 16), (True, 6)])
def test_warp_forward(pad, cl, ch):
    from dg_tta_amd import ops
    from oracle import tta as otta
    r, _ = _theta(2, 1, 0.08)
    torch.manual_seed(2)
    x = torch.randn(2, ch, 12, 14, 18)
    ref = otta.warp(x, r, pad)
    xd = x.to(DEV)
    if cl:
        xd = xd.contiguous(memory_format=torch.channels_last_3d)
    out = ops.affine_warp(xd, r.to(DEV), padding_mode=pad, tta_grid_algebra=True)
    _close(out, ref, atol=2e-5, what=f"warp {pad} cl={cl}")


def test_warp_backward_matches_autograd():
    from dg_tta_amd import ops
    from oracle import tta as otta
    _, rinv = _theta(1, 4, 0.08)
    torch.manual_seed(5)
    x = torch.randn(1, 8, 10, 12, 14, requires_grad=True)
    gy = torch.randn(1, 8, 10, 12, 14)
    otta.warp(x, rinv, "zeros").backward(gy)
    for cl in (False, True):
        xd = x.detach().to(DEV)
        if cl:
            xd = xd.contiguous(memory_format=torch.channels_last_3d)
        xd.requires_grad_(True)
        ops.affine_warp(xd, rinv.to(DEV), padding_mode="zeros", tta_grid_algebra=True).backward(gy.to(DEV))
        _close(xd.grad, x.grad, atol=5e-5, what=f"warp bwd cl={cl}")


@pytest.mark.parametrize("case", ["strong", "resize", "odd_channels", "singular", "minify", "border"])
def test_warp_backward_general_maps(case):
    """adjoint of the sampler for maps beyond the near-identity TTA case: owner-computes gather path, the atomic
    fallback it defers to (singular / strongly minifying maps) and border padding; torch fp32 autograd reference."""
    import torch.nn.functional as F
    from dg_tta_amd import ops
    torch.manual_seed(11)
    c, src, dst, pad = 8, (10, 12, 14), (10, 12, 14), "zeros"
    theta = torch.eye(3, 4)[None] + 0.3 * torch.randn(1, 3, 4)
    if case == "resize":
        dst = (7, 16, 9)
    elif case == "odd_channels":
        c = 5
    elif case == "singular":
        theta[0, 1] = 0.0
    elif case == "minify":
        theta = torch.eye(3, 4)[None] * 0.05
    elif case == "border":
        pad = "border"
    x = torch.randn(1, c, *src, requires_grad=True)
    gy = torch.randn(1, c, *dst)
    grid = F.affine_grid(theta, [1, c, *dst], align_corners=False)
    F.grid_sample(x, grid, mode="bilinear", padding_mode=pad, align_corners=False).backward(gy)
    for cl in (False, True):
        xd = x.detach().to(DEV)
        if cl:
            xd = xd.contiguous(memory_format=torch.channels_last_3d)
        xd.requires_grad_(True)
        ops.affine_warp(xd, theta.to(DEV), out_size=dst, padding_mode=pad).backward(gy.to(DEV))
        _close(xd.grad, x.grad, atol=1e-4, what=f"warp bwd {case} cl={cl}")


def test_get_batch_golden():
    from dg_tta_amd.tta.torch_utils import get_batch
    g = load_golden("get_batch")
    torch.manual_seed(21)      # the reference draws torch.rand(3) on the CPU generator
    imgs, lbls = get_batch([g["data"]], [0], [16, 16, 16], device=DEV)
    _close(imgs[0], g["img"], atol=3e-3, what="get_batch img")          # values ~ -300 +- 100: 1e-5 relative
    assert torch.equal(lbls[0].cpu(), g["lbl"])                          # integer label patch: bit exact
    imgs, lbls = get_batch([g["data"]], [0], [16, 16, 16], fixed_patch_idx="center", device=DEV)
    _close(imgs[0], g["cimg"], atol=3e-3, what="get_batch center img")
    assert torch.equal(lbls[0].cpu(), g["clbl"])
    torch.manual_seed(22)
    imgs, lbls = get_batch([g["small"]], [0], [16, 16, 16], device=DEV)
    assert lbls[0] is None
    _close(imgs[0], g["small_img"], atol=1e-4, what="get_batch small img")


# ------------------------------------------------------------------------------------------------ loss
def test_consistency_loss_golden_and_grad():
    from dg_tta_amd import ops
    from oracle import tta as otta
    g = load_golden("loss")
    ta, tb = g["ta"].clone().requires_grad_(True), g["tb"].clone().requires_grad_(True)
    ref = otta.consistency_loss(ta, tb)
    assert abs(float(ref) - float(g["loss"])) < 1e-6
    (ref / 16).backward()
    da, db = g["ta"].to(DEV).requires_grad_(True), g["tb"].to(DEV).requires_grad_(True)
    loss, dice = ops.consistency_loss(da, db)
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    (loss / 16).backward()
    _close(da.grad, ta.grad, atol=1e-8, rtol=1e-3, what="dloss/dlogits_a")
    _close(db.grad, tb.grad, atol=1e-8, rtol=1e-3, what="dloss/dlogits_b")


def test_consistency_loss_edge_cases():
    from dg_tta_amd import ops
    from oracle import tta as otta
    # all voxels masked out (channel sums <= 0) -> denominator.sum()==0 -> dice = 1 -> loss = 0 (torch_utils.py:97-98)
    neg = -torch.rand(1, 3, 4, 5, 6) - 0.1
    loss, dice = ops.consistency_loss(neg.to(DEV), neg.to(DEV))
    assert float(loss) == 0.0 and torch.equal(dice.cpu(), torch.ones(1, 3))
    assert float(otta.consistency_loss(neg, neg)) == 0.0
    # identical, unmasked inputs -> dice 1
    pos = torch.rand(2, 5, 6, 7, 9) + 0.1
    loss, dice = ops.consistency_loss(pos.to(DEV), pos.to(DEV))
    assert abs(float(loss)) < 1e-6
    # many classes, B=2, ragged volume
    torch.manual_seed(1)
    a, b = torch.randn(2, 21, 5, 7, 11), torch.randn(2, 21, 5, 7, 11)
    loss, _ = ops.consistency_loss(a.to(DEV), b.to(DEV))
    assert abs(float(loss) - float(otta.consistency_loss(a, b))) < 2e-6


def test_soft_dice_loss_standalone_golden_and_grad():
    """The drop-in `soft_dice_loss(a, b) -> [B,C]` (torch_utils.py:90-104) against the reference's own outputs
    (tests/golden/loss.npz: d_ab, d_aa, and the all-zero guard d_zz), in both memory layouts, with autograd."""
    from dg_tta_amd.tta.torch_utils import soft_dice_loss
    from oracle import tta as otta
    g = load_golden("loss")
    a, b = g["a"].to(DEV), g["b"].to(DEV)
    _close(soft_dice_loss(a, b), g["d_ab"], atol=2e-6, what="soft_dice a,b")
    _close(soft_dice_loss(a, a), g["d_aa"], atol=2e-6, what="soft_dice a,a")
    z = torch.zeros(1, 3, 4, 4, 4, device=DEV)
    assert torch.equal(soft_dice_loss(z, z).cpu(), g["d_zz"]) and torch.equal(g["d_zz"], torch.ones(1, 3))
    cl = a.contiguous(memory_format=torch.channels_last_3d), b.contiguous(memory_format=torch.channels_last_3d)
    _close(soft_dice_loss(*cl), g["d_ab"], atol=2e-6, what="soft_dice channels-last")
    # gradient of the loss as tta.py:269 forms it, against torch autograd on the oracle
    ra, rb = g["a"].clone().requires_grad_(True), g["b"].clone().requires_grad_(True)
    (1 - otta.soft_dice_loss(ra, rb)[:, 1:].mean()).backward()
    for layout in (torch.contiguous_format, torch.channels_last_3d):
        da = g["a"].to(DEV).contiguous(memory_format=layout).requires_grad_(True)
        db = g["b"].to(DEV).contiguous(memory_format=layout).requires_grad_(True)
        (1 - soft_dice_loss(da, db)[:, 1:].mean()).backward()
        _close(da.grad, ra.grad, atol=1e-9, rtol=1e-4, what="d soft_dice / da")
        _close(db.grad, rb.grad, atol=1e-9, rtol=1e-4, what="d soft_dice / db")
    # a larger ragged volume, many classes
    torch.manual_seed(4)
    x, y = torch.rand(2, 21, 9, 17, 33), torch.rand(2, 21, 9, 17, 33)
    _close(soft_dice_loss(x.to(DEV), y.to(DEV)), otta.soft_dice_loss(x, y), atol=2e-6, what="soft_dice ragged")


def test_argmax_dice_bit_exact():
    from dg_tta_amd import ops
    from dg_tta_amd.tta.torch_utils import dice_coeff
    from oracle import tta as otta
    g = load_golden("loss")
    torch.manual_seed(3)
    logits = torch.randn(1, 4, 8, 8, 8)
    am, counts = ops.argmax_dice(logits.to(DEV), g["dc_lab"].to(DEV))
    assert torch.equal(am.cpu(), logits.argmax(1))
    d = dice_coeff(am, g["dc_lab"].to(DEV), 4)
    assert torch.allclose(d.cpu(), otta.dice_coeff(logits.argmax(1), g["dc_lab"], 4), atol=1e-6)
    d2 = dice_coeff(g["dc_out"].to(DEV), g["dc_lab"].to(DEV), 4)
    assert torch.allclose(d2.cpu(), g["dc"], atol=1e-6)


# ------------------------------------------------------------------------------------------------ AdamW
def test_adamw_golden():
    from dg_tta_amd.optim import HipAdamW
    g = load_golden("adamw")
    p = torch.nn.Parameter(g["p0"].to(DEV))
    frozen = torch.nn.Parameter(torch.ones(5, device=DEV))        # grad None -> untouched (no decay either)
    opt = HipAdamW([p, frozen], lr=float(g["lr"]))
    for grad in (g["g1"], g["g2"]):
        p.grad = grad.to(DEV)
        opt.step()
        opt.zero_grad()
    _close(p, g["p2"], atol=1e-6, what="adamw 2 steps")
    assert torch.equal(frozen.detach().cpu(), torch.ones(5))
    assert p.grad is None


def test_adamw_overflow_guard_skips_the_step_and_halves_the_loss_scale():
    """fp16 storage path: one inf in a gradient must not reach the parameters or the Adam state; the step is
    skipped on the device, the next resolve_overflow() halves the scale and rolls the step counter back; the following
    clean step is the FIRST Adam step of the unscaled problem."""
    from dg_tta_amd.optim import HipAdamW
    torch.manual_seed(3)
    p = torch.nn.Parameter(torch.randn(5000, device=DEV))
    q = torch.nn.Parameter(torch.randn(300, device=DEV))
    gp, gq = torch.randn(5000, device=DEV), torch.randn(300, device=DEV)
    ref_p, ref_q = torch.nn.Parameter(p.detach().clone()), torch.nn.Parameter(q.detach().clone())
    opt = HipAdamW([p, q], lr=1e-2, grad_scale=1024.0)
    before = (p.detach().clone(), q.detach().clone())
    p.grad, q.grad = gp * 1024.0, gq * 1024.0
    q.grad[17] = float("inf")                     # the overflow sits in the OTHER tensor: all or nothing
    opt.step()
    assert torch.equal(p.detach(), before[0]) and torch.equal(q.detach(), before[1])
    assert torch.equal(opt.state[p]["exp_avg"], torch.zeros_like(p))
    assert opt.resolve_overflow() is True and opt.grad_scale == 512.0 and opt.skipped_steps == 1
    assert opt.state[p]["step"] == 0 and opt.state[q]["step"] == 0
    assert opt.resolve_overflow() is False        # settled
    p.grad, q.grad = gp * 512.0, gq * 512.0
    opt.step()
    assert opt.resolve_overflow() is False and opt.state[p]["step"] == 1
    ref = torch.optim.AdamW([ref_p, ref_q], lr=1e-2)
    ref_p.grad, ref_q.grad = gp.clone(), gq.clone()
    ref.step()
    _close(p, ref_p.detach().cpu(), atol=1e-6, what="first clean step after a skipped one")
    _close(q, ref_q.detach().cpu(), atol=1e-6, what="first clean step after a skipped one (q)")
    nan = HipAdamW([p], lr=1e-2, grad_scale=2.0, min_grad_scale=2.0)
    p.grad = torch.full_like(p, float("nan"))
    keep = p.detach().clone()
    nan.step()
    assert torch.equal(p.detach(), keep) and nan.resolve_overflow() and nan.grad_scale == 2.0     # floor


# ------------------------------------------------------------------------------------------------ network
def _models(cfg, seed=0):
    from oracle import unet as ounet
    from dg_tta_amd.unet import HipPlainConvUNet
    om = ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(cfg), seed), seed + 1)
    hm = HipPlainConvUNet(cfg, conv_impl=1)
    hm.load_state_dict(om.state_dict())
    return om, hm.to(DEV)


def _grad_check(om, hm, x, sel=None, atol_rel=2e-4):
    y_ref = om(x)
    if sel is not None:
        y_ref = y_ref[:, sel]
        hm.set_selected_classes(sel)
    y = hm(x.to(DEV))
    _close(y, y_ref, atol=2e-4 * float(y_ref.abs().max()), what="unet logits")
    torch.manual_seed(11)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    y.backward(gy.to(DEV))
    ref = dict(om.named_parameters())
    worst = 0.0
    for name, p in hm.named_parameters():
        r = ref[name].grad
        if r is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None, name
        scale = float(r.abs().max())
        if name.endswith("conv.bias") and ".convs." in name:
            # bias in front of InstanceNorm: mathematically zero gradient, both sides hold rounding noise only
            assert float(p.grad.abs().max()) < 1e-3 * max(1.0, float(gy.abs().max()))
            continue
        err = float((p.grad.cpu() - r).abs().max())
        worst = max(worst, err / max(scale, 1e-12))
        assert err <= atol_rel * scale + 1e-7, f"{name}: grad err {err:.3e} vs scale {scale:.3e}"
    return worst


def test_unet_small_forward_backward():
    om, hm = _models(SMALL_CFG)
    torch.manual_seed(2)
    x = torch.randn(1, 12, 16, 16, 16)
    _grad_check(om, hm, x)


def test_unet_selected_rows_and_batch2():
    cfg = dict(SMALL_CFG, features=(6, 10, 14), num_classes=11)       # odd channel counts
    om, hm = _models(cfg, seed=3)
    torch.manual_seed(4)
    x = torch.randn(2, 12, 8, 16, 24)
    _grad_check(om, hm, x, sel=torch.tensor([0, 3, 4, 9]))


def test_unet_frozen_and_norm_only_grads():
    from dg_tta_amd.tta.torch_utils import fix_all, release_norms
    om, hm = _models(SMALL_CFG)
    x = torch.randn(1, 12, 16, 16, 16)
    hm.apply(fix_all)
    y = hm(x.to(DEV))
    assert not y.requires_grad
    hm.apply(release_norms)
    om.apply(fix_all)
    for m in om.modules():
        if isinstance(m, torch.nn.InstanceNorm3d):
            for p in m.parameters():
                p.requires_grad_(True)
    y = hm(x.to(DEV))
    y.square().mean().backward()
    om(x).square().mean().backward()
    for (n, p), (_, q) in zip(hm.named_parameters(), om.named_parameters()):
        if ".norm." in n:
            _close(p.grad, q.grad, atol=2e-4 * float(q.grad.abs().max()) + 1e-9, what=n)
        else:
            assert p.grad is None


def test_cpu_tensor_is_rejected():
    from dg_tta_amd import ops
    from dg_tta_amd._lib import DgttaError
    with pytest.raises(DgttaError):
        ops.mind3d(torch.zeros(1, 1, 8, 8, 8), torch.zeros(1, 12, 8, 8, 8))


def test_unet_inplace_grad_accumulation_matches_autograd_path():
    om, hm = _models(SMALL_CFG)
    torch.manual_seed(6)
    x = torch.randn(1, 12, 16, 16, 16).to(DEV)
    gy = torch.randn(1, 9, 16, 16, 16).to(DEV)
    for _ in range(2):                       # two accumulation steps through autograd
        hm(x).backward(gy)
    ref = {n: p.grad.clone() for n, p in hm.named_parameters() if p.grad is not None}
    hm.zero_grad()
    hm.accumulate_grads_in_place = True
    for _ in range(2):
        hm(x).backward(gy)
    for n, p in hm.named_parameters():
        if n in ref:
            assert torch.allclose(p.grad, ref[n], rtol=1e-5, atol=1e-6 * float(ref[n].abs().max()) + 1e-9), n
        else:
            assert p.grad is None


@pytest.mark.parametrize("shape", [(2, 16, 20, 22, 37), (1, 32, 9, 17, 70), (3, 16, 16, 16, 16)])
def test_owner_computes_warp_backward_matches_autograd_on_ragged_tiles(shape):
    """warp_bwd_gather_kernel (one lane owns one grad_src voxel, atomic-free, fixed summation order) against autograd through
    the oracle's sampler (tta.py:572-575) on ragged tiles, 2 channel blocks and a stronger map (many candidates per voxel);
    deterministic: two runs are torch.equal.  (Round 3's cooperative owner / loader prototype, which this test used to pin
    bit for bit against this kernel, was measured no faster and removed in round 5.)"""
    from dg_tta_amd import ops
    from oracle import tta as otta
    b, c, d, h, w = shape
    torch.manual_seed(sum(shape))
    gy = torch.randn(b, c, d, h, w, device=DEV).contiguous(memory_format=torch.channels_last_3d)
    thetas = [_theta(b, 3, 0.08)[1], torch.eye(3, 4)[None].repeat(b, 1, 1) * 0.45 + 0.02 * torch.randn(b, 3, 4)]
    for theta in thetas:
        outs = []
        for _ in range(2):
            x = torch.zeros(b, c, d, h, w, device=DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
            ops.affine_warp(x, theta.to(DEV), padding_mode="zeros", tta_grid_algebra=True).backward(gy)
            outs.append(x.grad.clone())
        assert torch.equal(outs[0], outs[1])
        assert float(outs[0].abs().sum()) > 0
        xr = torch.zeros(b, c, d, h, w, requires_grad=True)
        otta.warp(xr, theta, "zeros").backward(gy.cpu().contiguous())
        _close(outs[0], xr.grad, atol=5e-5, what="warp bwd vs autograd")


def test_consistency_loss_16_class_kernels_match_the_oracle_and_the_generic_kernels(monkeypatch):
    """Round 3: softdice_fwd16 / bwd16 (four lanes per voxel, softmax in registers) for the plan's C_opt = 16 against the
    CPU oracle (tta.py:263-269 restated) and against the generic LDS-tile kernels (DGTTA_SOFTDICE16=0): loss, per-class Dice
    and both gradients; voxels whose logits are all zero (outside the warped field of view) and whose class sum is negative
    are masked out in both; the pair form (one batched tensor) is covered as well."""
    from conftest import reload_kernel_switches
    from dg_tta_amd import ops
    from oracle import tta as otta
    torch.manual_seed(8)
    a, b = torch.randn(4, 16, 9, 13, 22) * 3, torch.randn(4, 16, 9, 13, 22) * 3
    a[:, :, :2] = 0.0                       # zero-padded border of the inverse warp
    b[:, :, :, :3] = 0.0
    a[1, :, 4] -= 5.0                        # negative class sums: masked
    ta, tb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = otta.consistency_loss(ta, tb)
    (ref * 3.0).backward()
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DGTTA_SOFTDICE16", flag)
        reload_kernel_switches()
        da = a.to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        db = b.to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        loss, dice = ops.consistency_loss(da, db)
        (loss * 3.0).backward()
        outs[flag] = (float(loss), dice.cpu(), da.grad.cpu(), db.grad.cpu())
        assert abs(float(loss) - float(ref)) < 2e-6
        _close(da.grad, ta.grad, atol=1e-9, rtol=1e-3, what=f"dloss/da ({flag})")
        _close(db.grad, tb.grad, atol=1e-9, rtol=1e-3, what=f"dloss/db ({flag})")
    l1, d1, ga1, gb1 = outs["1"]
    l0, d0, ga0, gb0 = outs["0"]
    assert abs(l1 - l0) < 5e-7 and float((d1 - d0).abs().max()) < 2e-6
    scale = float(ga0.abs().max())
    assert float((ga1 - ga0).abs().max()) < 2e-5 * scale and float((gb1 - gb0).abs().max()) < 2e-5 * scale
    # pair form: one batched tensor [2B, 16, ...], first half = branch a
    monkeypatch.setenv("DGTTA_SOFTDICE16", "1")
    reload_kernel_switches()
    both = torch.cat([a, b]).to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    pa, pb = both[:4], both[4:]
    pa._dgtta_pair = pb._dgtta_pair = both
    loss, _ = ops.consistency_loss(pa, pb)
    (loss * 3.0).backward()
    assert abs(float(loss) - l1) < 1e-7
    assert torch.equal(both.grad[:4].cpu(), ga1) and torch.equal(both.grad[4:].cpu(), gb1)
