"""Test infrastructure: drives the PRODUCT loop with the draw stream of the CPU oracle / the reference.

The reference runs on one device, so on the CPU every draw (sample choice, patch offsets, GIN kernels, affine matrices,
MIND noise) comes from the ONE CPU generator in call order (dg_tta/tta/tta.py:221-275, gin.py:193-195, mind.py:150).  The
product draws GIN alpha / MIND noise on the device generator, as the reference does on a GPU.  To compare a product run
with a CPU run of the oracle or with a reference-generated fixture, device draws are re-routed through the CPU generator
for the duration of the context.  Only tests/ and bench.py's parity legs use this (never the timed or shipped path)."""
import contextlib

import torch


@contextlib.contextmanager
def cpu_rng_for_device_draws():
    real_rand, real_randn = torch.rand, torch.randn

    def rand(*a, **k):
        dev = k.pop("device", None)
        t = real_rand(*a, **k)
        return t.to(dev) if dev is not None else t

    def randn(*a, **k):
        dev = k.pop("device", None)
        t = real_randn(*a, **k)
        return t.to(dev) if dev is not None else t

    from dg_tta_amd import mind as hmind
    real_draw = hmind.draw_noise_

    def draw_noise_(slot):          # the batched path draws a branch's MIND noise in place into its slot of the batch
        return slot.copy_(real_randn(slot.shape))

    torch.rand, torch.randn = rand, randn
    hmind.draw_noise_ = draw_noise_
    try:
        yield
    finally:
        torch.rand, torch.randn = real_rand, real_randn
        hmind.draw_noise_ = real_draw
