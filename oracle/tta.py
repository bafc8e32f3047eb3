"""TTA inner loop, CPU restatement (oracle; test infrastructure).

Follows /root/reference/dg_tta/tta/tta.py:190-281 (epoch / accumulation loop),
:480-579 (calc_branch), dg_tta/tta/torch_utils.py:13-117,214-263 (get_batch, soft_dice_loss,
dice_coeff, map_label, generate_label_mapping, get_map_idxs),
dg_tta/tta/augmentation_utils.py:156-170 (get_rand_affine).
Random draws are explicit arguments where a HIP kernel must be fed the same numbers.
"""
import torch
import torch.nn.functional as F

from . import gin as ogin
from . import mind as omind

START_CLASS = 1     # tta.py:103


# ---------------------------------------------------------------- label mapping (torch_utils.py:230-263)
def generate_label_mapping(src, tgt):
    assert all(isinstance(k, str) for k in src) and all(isinstance(k, str) for k in tgt)
    assert set(src).intersection(tgt), "There are no intersecting label names in given dicts."
    out = {}
    for k in list(src.keys()) + list(tgt.keys()):
        if k in src and k in tgt and k not in out:
            out[k] = (src[k], tgt[k])
    return out


def get_map_idxs(label_mapping, optimized_labels, input_type):
    assert input_type in ("pretrain_labels", "tta_labels")
    assert optimized_labels[0] == "background"
    col = 0 if input_type == "pretrain_labels" else 1
    return torch.as_tensor([label_mapping[l][col] for l in optimized_labels])


def map_label(label, map_idxs, input_format):
    assert input_format in ("logits", "argmaxed")
    if input_format == "logits":
        return label[:, map_idxs]                    # == transpose(0,1)[idx].transpose(0,1) (torch_utils.py:221)
    out = torch.zeros_like(label)
    for new, old in enumerate(map_idxs):
        out[label == old] = new
    return out


# ---------------------------------------------------------------- metrics (torch_utils.py:90-117)
def soft_dice_loss(a, b):
    bsz = a.shape[0]
    v = a.shape[2] * a.shape[3] * a.shape[4]
    nom = (2.0 * a * b).reshape(bsz, -1, v).mean(2)
    den = 0.5 * ((a + b) ** 2).reshape(bsz, -1, v).mean(2)
    if den.sum() == 0.0:
        return nom * 0.0 + 1.0
    return nom / den


def consistency_loss(ta, tb):
    """tta.py:263-269: masked softmax of both branches, 1 - mean soft dice over classes >= 1."""
    mask = (ta.sum(1, keepdim=True) > 0.0).float() * (tb.sum(1, keepdim=True) > 0.0).float()
    sa = ta.softmax(1) * mask
    sb = tb.softmax(1) * mask
    return 1 - soft_dice_loss(sa, sb)[:, START_CLASS:].mean()


def dice_coeff(outputs, labels, max_label):
    d = torch.zeros(max_label - 1)
    for l in range(1, max_label):
        i = (outputs == l).view(-1).float()
        t = (labels == l).view(-1).float()
        d[l - 1] = (2.0 * torch.mean(i * t)) / (1e-8 + torch.mean(i) + torch.mean(t))
    return d


# ---------------------------------------------------------------- sampling (torch_utils.py:13-82)
def patch_affine(vol_shape, patch_size, rand3=None):
    """The 3x4 matrix get_batch builds: diag(P/S flipped) + offset column. rand3 = torch.rand(3) or None (center)."""
    p = torch.as_tensor(patch_size)
    s = torch.as_tensor(tuple(vol_shape))
    scales = torch.cat([(p / s).flip(0), torch.tensor([1.0])])
    aff = scales.diag()
    if rand3 is not None:
        off = (2.0 * rand3 - 1.0) * ((s - p) / s).clip(min=0.0)
        aff[:, -1] = torch.cat([off.flip(0), torch.tensor([1.0])])
    return aff[:3]


def get_batch_item(data, patch_size, rand3=None):
    """One element of get_batch: data [1+K,Dv,Hv,Wv] -> (img [1,1,P], label [1,1,P] int64 or None)."""
    theta = patch_affine(data.shape[-3:], patch_size, rand3)[None]
    grid = F.affine_grid(theta, (1, 1, *patch_size), align_corners=False)
    mn = data[0].min()
    img = F.grid_sample(data[0][None, None] - mn, grid, align_corners=False, padding_mode="zeros") + mn
    if data[1:].numel() == 0:
        return img, None
    lbl = F.grid_sample(data[1:][None], grid, align_corners=False, padding_mode="zeros", mode="nearest")
    with_bg = torch.cat([(lbl.sum(1, keepdim=True) < 1.0).float(), lbl], dim=1)
    return img, with_bg.argmax(1, keepdim=True)


# ---------------------------------------------------------------- spatial augmentation
def rand_affine_from_draw(randn_b34, strength=0.05):
    """augmentation_utils.py:156-170 with the randn(B,3,4) draw passed in (flip=False)."""
    b = randn_b34.shape[0]
    aff = torch.cat((randn_b34 * strength + torch.eye(3, 4).unsqueeze(0),
                     torch.tensor([0, 0, 0, 1]).view(1, 1, 4).repeat(b, 1, 1)), 1)
    return aff[:, :3], aff.inverse()[:, :3]


def warp(x, theta, padding_mode):
    """tta.py:523-551 / :572-575: grid = (affine_grid(theta) - id) + id, then grid_sample (bilinear)."""
    b = x.shape[0]
    size = [b, 1] + list(x.shape[2:])
    ident = F.affine_grid(torch.eye(4)[:3][None].repeat(b, 1, 1), size, align_corners=False)
    grid = 0.0 * ident + (F.affine_grid(theta, size, align_corners=False) - ident)
    grid = grid + ident
    return F.grid_sample(x, grid, padding_mode=padding_mode, align_corners=False)


# ---------------------------------------------------------------- branch / step (tta.py:221-279, 480-579)
def calc_branch(model, imgs, map_idxs, gin_draw=None, affine_draw=None, mind_noise=None):
    """One branch: [GIN] -> [affine warp, border] -> MIND pre-hook -> model -> map_label -> [inverse warp, zeros].

    gin_draw = (alpha, ks, kers, shifts) or None; affine_draw = randn(B,3,4) or None;
    mind_noise = randn(B,12,D,H,W) (the device randn_like of mind.py:150) or None for a 1-channel net.
    """
    x = imgs
    if gin_draw is not None:
        x = ogin.gin_chain(x, *gin_draw)
    if affine_draw is not None:
        r, r_inv = rand_affine_from_draw(affine_draw)
        x = warp(x, r, "border")
    if mind_noise is not None:
        x = omind.mind3d(x, mind_noise)
    y = map_label(model(x), map_idxs, "logits")
    if affine_draw is not None:
        y = warp(y, r_inv, "zeros")
    return y


def tta_step(model, imgs, map_idxs, draws_a, draws_b, accum, backward=True):
    """One accumulation step. draws_* = dict(gin_draw=, affine_draw=, mind_noise=). Returns loss (detached)."""
    ta = calc_branch(model, imgs, map_idxs, **draws_a)
    tb = calc_branch(model, imgs, map_idxs, **draws_b)
    loss = consistency_loss(ta, tb)
    if backward:
        (loss / accum).backward()
    return loss.detach()


def adamw_reference_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor math (decoupled weight decay, bias-corrected), tta.py:185,278."""
    p = p * (1 - lr * wd)
    m = m + (1 - b1) * (g - m)          # lerp form used by torch
    v = v * b2 + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# ---------------------------------------------------------------- whole unit (tta.py:189-340)
def draw_branch(batch, patch_size, gin=True, affine=True, mind=True):
    """The draws of ONE calc_branch call in the reference's order on the CPU generator: GIN (gin.py:193-195, 65-103),
    affine (augmentation_utils.py:158), MIND noise (mind.py:150, inside the model's pre-hook)."""
    return dict(gin_draw=ogin.draw_gin_params(batch) if gin else None,
                affine_draw=torch.randn(batch, 3, 4) if affine else None,
                mind_noise=torch.randn(batch, 12, *patch_size) if mind else None)


def tta_unit(model, optimizer, data_list, patch_size, map_pre, map_tta, epochs, start, accum, eval_patches=1, batch=1):
    """Epoch / accumulation loop of tta.py:189-340 for one (sample, ensemble) unit with params_with_grad='all',
    GIN + affine in both branches and a MIND net.  Every draw comes from the global CPU generators in the reference's
    order: np.random.choice + rand(3) (get_batch), branch a, branch b, ...; eval: np.random.choice, MIND noise.
    Returns (tta_losses[epochs], eval_dices[epochs], step_losses)."""
    import numpy as np
    n_opt = len(map_pre)
    tta_losses, eval_dices, all_steps = torch.zeros(epochs), torch.zeros(epochs), []
    for p in model.parameters():
        p.requires_grad_(False)
    for epoch in range(epochs):
        model.train()
        if epoch == start:
            for p in model.parameters():
                p.requires_grad_(True)
        step_losses = []
        for _ in range(accum):
            idxs = np.random.choice(range(len(data_list)), batch).tolist()
            imgs = torch.cat([get_batch_item(data_list[i], patch_size, torch.rand(3))[0] for i in idxs], dim=0)
            da = draw_branch(batch, patch_size)
            ta = calc_branch(model, imgs, map_pre, **da)
            db = draw_branch(batch, patch_size)
            tb = calc_branch(model, imgs, map_pre, **db)
            loss = consistency_loss(ta, tb)
            step_losses.append(loss.detach())
            if epoch >= start:
                (loss / accum).backward()
        if epoch >= start:
            optimizer.step()
            optimizer.zero_grad()
        tta_losses[epoch] = torch.stack(step_losses).mean().item()
        all_steps += step_losses
        with torch.inference_mode():
            model.eval()
            for _ in range(eval_patches):
                idxs = np.random.choice(range(len(data_list)), batch).tolist()
                items = [get_batch_item(data_list[i], patch_size, None) for i in idxs]
                keep = [(im, lb) for im, lb in items if lb is not None]
                if not keep:
                    eval_dices[epoch] = float("nan")
                    continue
                imgs = torch.cat([im for im, _ in keep], dim=0)
                labels = torch.cat([lb for _, lb in keep], dim=0)
                out = map_label(model(omind.mind3d(imgs, torch.randn(imgs.shape[0], 12, *patch_size))), map_pre, "logits")
                labels = map_label(labels, map_tta, "argmaxed").long()
                eval_dices[epoch] += 1 / eval_patches * dice_coeff(out.argmax(1), labels, n_opt).nanmean().item()
    return tta_losses, eval_dices, torch.stack(all_steps)
