"""Source-domain pre-training THROUGH this engine: what nnU-Net's training loop does around the DG
trainers' network, reduced to its arithmetic - random patches, the trainer's forward pre-hooks (`gin_hook` with internal
augmentation on, `mind_hook`: dg_tta/pretraining/nnUNetTrainer_GIN_MIND.py:55-57), nnU-Net's DC_and_CE_loss
(csrc/dice_ce.hip), backward through the HIP network, AdamW (csrc/adamw.hip).  nnU-Net's own loop (poly-lr SGD, deep
supervision, its augmentation pipeline) stays out of scope; this exists so that bench.py and the tests can start TTA from
TRAINED weights - the TS104 checkpoints cannot be downloaded here (config_log_utils.py:307-350) - instead of He-initialised
ones, on which every Dice figure is ~0."""
import torch

from .. import ops
from ..optim import HipAdamW
from ..tta.torch_utils import get_batch
from ..utils import disable_internal_augmentation, enable_internal_augmentation, numpy_rng


def pretrain_supervised(net, cases, patch_size, label_to_class, steps, batch=2, lr=3e-3, device="cuda", internal_gin=True,
                        log_every=0, optimizer=None):
    """Trains `net` (a HipPlainConvUNet with its DG pre-hooks registered) on `cases` (list of [1+K, X, Y, Z] tensors: image +
    one-hot label channels, the layout get_batch reads).  label_to_class: int64 tensor, dataset label id (0 = background,
    i = label channel i) -> class index of the network output (pretrain class ids; with `net.selected_classes` set:
    positions in the selection).  Draws come from the thread's generators (utils.rng_scope / the global ones) in a fixed
    order, so a seed reproduces the weights.  `optimizer`: continue with this HipAdamW (its moments) instead of a fresh one.
    Returns the per-step losses (one device->host copy at the end)."""
    device = torch.device(device)
    opt = optimizer if optimizer is not None else HipAdamW([p for p in net.parameters()], lr=lr, weight_decay=0.0,
                                                            grad_scale=getattr(net, "loss_scale", 1.0))
    lut = label_to_class.to(device)
    was_training = net.training
    net.train()
    for p in net.parameters():
        p.requires_grad_(True)
    (enable_internal_augmentation if internal_gin else disable_internal_augmentation)()
    losses = []
    try:
        for step in range(steps):
            if opt.resolve_overflow():
                print(f"  pre-training: gradient overflow, step skipped, loss scale -> {opt.grad_scale:g}")
            idxs = numpy_rng().choice(range(len(cases)), batch).tolist()
            with torch.no_grad():
                imgs, labels = get_batch(cases, idxs, patch_size, fixed_patch_idx=None, device=device)
            imgs = torch.cat(imgs, dim=0)
            target = lut[torch.cat(labels, dim=0)[:, 0]]
            loss, _, _ = ops.dice_ce_loss(net(imgs), target)
            torch.autograd.backward(loss, grad_tensors=torch.full((), float(opt.grad_scale), dtype=torch.float32, device=device))
            opt.step()
            opt.zero_grad(set_to_none=True)
            losses.append(loss.detach())
            if log_every and (step % log_every == 0 or step == steps - 1):
                print(f"  pre-training step {step:4d}: loss {float(loss):.4f}", flush=True)
    finally:
        disable_internal_augmentation()
        net.train(was_training)
    return torch.stack(losses).cpu()
