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
