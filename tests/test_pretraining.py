"""Pre-training side: the MultiRes trainers' discrete low-resolution transform on the GPU against the
scipy restatement with the reference's numpy draw order, and the trainers' forward pre-hooks on a batch of 2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("orders", [(0, 3), (1, 0)])       # (down, up): the MultiRes trainers' setting / the function defaults
def test_discrete_low_resolution_transform_matches_oracle(orders):
    from dg_tta_amd.pretraining import SimulateDiscreteLowResolutionTransform
    from oracle import discrete_downsampling as od
    rng = np.random.default_rng(3)
    data = rng.normal(0, 1, (3, 2, 24, 30, 36)).astype(np.float32)       # [batch, channels, X, Y, Z]
    kw = dict(zoom_range=(1 / 6, 1 / 4, 1 / 2), zoom_axes_invidually=True, order_downsample=orders[0],
              order_upsample=orders[1], ignore_axes=None)
    np.random.seed(11)
    ref = od.transform(data.copy(), p_per_sample=0.5, p=1.0, **kw)
    np.random.seed(11)
    tr = SimulateDiscreteLowResolutionTransform(per_channel=False, p_per_channel=1.0, p_per_sample=0.5, **kw)
    out = tr(data=data.copy())["data"]
    assert out.shape == ref.shape and out.dtype == ref.dtype
    assert np.abs(out - ref).max() < 1e-5 * np.abs(ref).max()
    changed = [not np.array_equal(out[b], data[b]) for b in range(3)]
    assert any(changed) and not all(changed)                 # p_per_sample = 0.5 with this seed: some samples untouched
    # CUDA tensors in, CUDA tensors out (the GPU-resident path), same draws
    np.random.seed(11)
    t = [torch.from_numpy(data[b]).to(DEV) for b in range(3)]
    out_t = tr(data=t)["data"]
    for b in range(3):
        assert out_t[b].is_cuda and np.abs(out_t[b].cpu().numpy() - ref[b]).max() < 1e-5 * np.abs(ref).max()
    # dummy-2D augmentation: axis 0 keeps its resolution
    np.random.seed(5)
    r2 = od.transform(data.copy(), p_per_sample=1, p=1.0, **{**kw, "ignore_axes": (0,)})
    np.random.seed(5)
    o2 = SimulateDiscreteLowResolutionTransform(p_per_channel=1.0, p_per_sample=1, **{**kw, "ignore_axes": (0,)})(data=data.copy())["data"]
    assert np.abs(o2 - r2).max() < 1e-5 * np.abs(r2).max()


def test_trainer_hooks_on_a_training_batch():
    """build_network_architecture of the GIN_MIND trainer: 12 input channels, hooks in the reference's order; with internal
    augmentation on, a batch of 2 gets one GIN chain per item (groups = nb) and then the MIND descriptor - checked against
    the oracle fed with the same draws."""
    import contextlib
    from conftest import SMALL_CFG
    from dg_tta_amd.gin import gin_hook
    from dg_tta_amd.mind import mind_hook
    from dg_tta_amd.pretraining import build_network_architecture
    from dg_tta_amd.utils import disable_internal_augmentation, get_internal_augmentation_enabled
    from oracle import gin as ogin, mind as omind, unet as ounet
    net = build_network_architecture(SMALL_CFG, "nnUNetTrainer_GIN_MIND_MultiRes").to(DEV)
    try:
        assert get_internal_augmentation_enabled()
        assert list(net._forward_pre_hooks.values()) == [gin_hook, mind_hook] and net.cfg["in_channels"] == 12
        om = ounet.perturb_affine(ounet.init_he(ounet.PlainConvUNetOracle(SMALL_CFG), 3), 4)
        net.load_state_dict(om.state_dict())
        torch.manual_seed(0)
        x = torch.randn(2, 1, 16, 16, 16)
        # same draws on both sides: alpha (device generator in the product) and kernels from the CPU stream, MIND noise
        real_rand, real_randn = torch.rand, torch.randn

        def rand(*a, **k):
            dev = k.pop("device", None)
            t = real_rand(*a, **k)
            return t.to(dev) if dev is not None else t

        def randn(*a, **k):
            dev = k.pop("device", None)
            t = real_randn(*a, **k)
            return t.to(dev) if dev is not None else t
        torch.rand, torch.randn = rand, randn
        try:
            torch.manual_seed(21)
            out = net(x.to(DEV))
        finally:
            torch.rand, torch.randn = real_rand, real_randn
        torch.manual_seed(21)
        alpha, ks, kers, shifts = ogin.draw_gin_params(2)
        noise = torch.randn(2, 12, 16, 16, 16)
        with torch.no_grad():
            ref = om(omind.mind3d(ogin.gin_chain(x, alpha, ks, kers, shifts), noise))
        assert (out.detach().cpu() - ref).abs().max() < 3e-4 * ref.abs().max()
        assert out.requires_grad                              # training: gradients flow through the HIP network
    finally:
        disable_internal_augmentation()


@pytest.mark.parametrize("case", [(2, 9, (10, 12, 14)), (1, 105, (8, 8, 40)), (3, 16, (16, 16, 16)), (1, 2, (5, 7, 67))])
def test_dice_ce_loss_matches_torch(case):
    """csrc/dice_ce.hip (nnU-Net's DC_and_CE_loss [3P]: per-sample soft Dice without background, smooth 1e-5, + cross-entropy)
    against a plain torch fp32 evaluation and its autograd: loss, its two parts, dice[B,C], the logit gradient under an
    upstream scale; ignored voxels (labels outside [0, C)), absent classes, channels-last and contiguous logits."""
    from dg_tta_amd import ops
    b, c, shp = case
    g = torch.Generator().manual_seed(b * 100 + c)
    logits = (torch.randn(b, c, *shp, generator=g) * 3).requires_grad_(True)
    labels = torch.randint(0, c, (b, 1, *shp), generator=g)
    labels[0, 0, 0, :3] = -1                       # ignored
    labels[0, 0, 1, :2] = c + 5                    # ignored
    if c > 4:
        labels[labels == c - 2] = 0                # an absent class
    valid = (labels >= 0) & (labels < c)
    safe = labels.clamp(0, c - 1)
    logp = torch.log_softmax(logits, 1)
    ce = -(logp.gather(1, safe) * valid).sum() / valid.sum()
    p = logits.softmax(1) * valid
    oh = torch.nn.functional.one_hot(safe[:, 0], c).permute(0, 4, 1, 2, 3).float() * valid
    dice = (2 * (p * oh).sum((2, 3, 4)) + 1e-5) / (p.sum((2, 3, 4)) + oh.sum((2, 3, 4)) + 1e-5)
    ref = ce - dice[:, 1:].mean()
    (ref * 0.37).backward()
    for fmt in (torch.channels_last_3d, torch.contiguous_format):
        x = logits.detach().to(DEV).contiguous(memory_format=fmt).requires_grad_(True)
        loss, d, parts = ops.dice_ce_loss(x, labels.to(DEV))
        (loss * 0.37).backward()
        assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
        assert abs(float(parts[0]) - float(ce)) < 2e-5 * max(1.0, float(ce))
        assert (d.cpu() - dice.detach()).abs().max() < 2e-5
        gerr = (x.grad.cpu() - logits.grad).abs().max() / logits.grad.abs().max()
        assert float(gerr) < 2e-4, float(gerr)
    # deterministic
    l2, _, _ = ops.dice_ce_loss(logits.detach().to(DEV), labels.to(DEV))
    l3, _, _ = ops.dice_ce_loss(logits.detach().to(DEV), labels.to(DEV))
    assert torch.equal(l2, l3)


@pytest.mark.parametrize("storage", [torch.float32, torch.float16])
def test_supervised_pretraining_learns_the_atlas_task(storage):
    """dg_tta_amd.pretraining.supervised: the small net, GIN + MIND hooks in their pre-training role, 400 steps on the source
    domain of the synthetic atlas task: the loss falls and an UNSEEN source case is segmented with hard Dice > 0.6 (the
    torch-CPU run of tests/golden/make_golden_r5.py reaches 0.81 after the same 400 steps); the same seed gives the same weights."""
    import numpy as np
    from conftest import SMALL_CFG
    from dg_tta_amd.mind import MIND3D
    from dg_tta_amd.pretraining.hooks import register_dg_hooks
    from dg_tta_amd.pretraining.supervised import pretrain_supervised
    from dg_tta_amd.synthetic import atlas_case, he_init_
    from dg_tta_amd.tta.torch_utils import dice_coeff, get_batch, release_resident
    from dg_tta_amd.unet import HipPlainConvUNet
    from dg_tta_amd.utils import disable_internal_augmentation
    cases = [atlas_case(24, 4, s, "source") for s in range(6)]
    lut = torch.tensor([0, 3, 8, 2, 5])                       # dataset label id -> pretrain class of the 9-class net

    def train(steps):
        net = he_init_(HipPlainConvUNet(SMALL_CFG, act_dtype=storage), seed=7)
        handles = register_dg_hooks(net, "nnUNetTrainer_GIN_MIND")
        net = net.to(DEV)
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        np.random.seed(5)
        losses = pretrain_supervised(net, cases, [16, 16, 16], lut, steps=steps, batch=4, lr=3e-3, device=DEV)
        for h in handles:
            h.remove()
        return net, losses
    try:
        net, losses = train(400)
        assert float(losses[-25:].mean()) < 0.6 * float(losses[:10].mean())
        test = atlas_case(24, 4, 77, "source")
        with torch.no_grad():
            net.eval()
            imgs, labels = get_batch([test], [0], [16, 16, 16], "center", DEV)
            torch.manual_seed(1)
            out = net.forward(MIND3D()(imgs[0], out_dtype=storage))
            pred = out.argmax(1, keepdim=True)
            d = dice_coeff(pred, lut.to(DEV)[labels[0]], 9)
        present = [1, 2, 4, 7]                                # classes 2, 3, 5, 8 (dice_coeff drops the background)
        print(f"\npre-trained {storage}: loss {float(losses[:10].mean()):.3f} -> {float(losses[-25:].mean()):.3f}, hard Dice of an unseen source case {d[present].tolist()}")
        assert float(d[present].mean()) > 0.6, d.tolist()
        if storage == torch.float32:
            net2, losses2 = train(12)
            assert torch.equal(losses2, losses[:12])
    finally:
        disable_internal_augmentation()
        release_resident()
