"""Patch sampler, losses, label mapping, grad freezing, hook helpers — same names and semantics as the reference's
dg_tta/tta/torch_utils.py; numeric work runs in the HIP kernels (sampler: csrc/warp.hip, loss: csrc/softdice.hip,
Dice counting: csrc/unet_ref.hip argmax_dice_kernel)."""
from collections import OrderedDict

import torch

from .. import ops
from ..utils import cpu_generator, upload_async

_VOLUME_CACHE = {}


def _resident(data, device):
    """Uploads a preprocessed case once and keeps it in HBM (the reference re-uploads it on every call,
    torch_utils.py:60).  Returns (image [1,1,Dv,Hv,Wv] fp32, min scalar [1], label map [1,1,Dv,Hv,Wv] fp32 or None).
    An entry is tied to its source tensor: a hit requires the SAME tensor object at the same version (an address can be
    reused by a later case of the same shape), and the device copy is dropped when the host tensor is collected or when
    tta_main releases the sample - never behind a running epoch's back."""
    import weakref
    key = (id(data), str(device))
    hit = _VOLUME_CACHE.get(key)
    if hit is not None and hit[0]() is data and hit[1] == data._version:
        return hit[2]
    img = data[0][None, None].float().contiguous().to(device)
    mn = data[0].min().reshape(1).float().to(device)                  # torch_utils.py:58
    lab = None
    if data[1:].numel() != 0:
        segs = data[1:][None]
        # get_argmaxed_segs (torch_utils.py:79-82) commutes with nearest sampling: apply it once to the volume
        lab = torch.cat([(segs.sum(1, keepdim=True) < 1.0).float(), segs.float()], dim=1).argmax(1, keepdim=True)
        lab = lab.float().contiguous().to(device)
    _VOLUME_CACHE[key] = (weakref.ref(data), data._version, (img, mn, lab))
    weakref.finalize(data, _VOLUME_CACHE.pop, key, None)
    return img, mn, lab


def release_resident(tensor_list=None):
    """Frees the device copies of cached volumes: those of `tensor_list` (tta_main, after each sample), or all of them.
    Entries are never evicted behind a running epoch's back: kernels enqueued on the input-pipeline stream may still
    read them."""
    if tensor_list is None:
        _VOLUME_CACHE.clear()
        return
    ids = {id(t) for t in tensor_list}
    for key in [k for k in _VOLUME_CACHE if k[0] in ids]:
        del _VOLUME_CACHE[key]


def get_batch(tensor_list, batch_idxs, patch_size, fixed_patch_idx=None, device="cuda"):
    """Reference: torch_utils.py:13-76.  Random (torch.rand(3) on the CPU generator) or centre patch via affine
    resampling; image trilinear with the (x - min) + min trick, labels nearest."""
    assert fixed_patch_idx in range(8) or fixed_patch_idx is None or fixed_patch_idx == "center"
    device = torch.device(device)
    b_img, b_label = [], []
    t_patch = torch.as_tensor(patch_size)
    t_shape = torch.as_tensor(tensor_list[0].shape[-3:])
    scales = torch.cat([(t_patch / t_shape).flip(0), torch.tensor([1.0])], dim=0)
    patch_affine = scales.diag()
    for b in range(len(batch_idxs)):
        data = tensor_list[batch_idxs[b]]
        if fixed_patch_idx != "center":
            rand_offset = 2.0 * torch.rand(3, generator=cpu_generator()) - 1.0
            offset_range = ((t_shape - t_patch) / t_shape).clip(min=0.0)
            ranged = rand_offset * offset_range
            patch_affine[:, -1] = torch.cat([ranged.flip(0), torch.tensor([1.0])], dim=0)
        theta = upload_async([patch_affine[:3][None].float().contiguous()], device)[0]
        img, mn, lab = _resident(data, device)
        b_img.append(ops.affine_sample(img, theta, patch_size, "zeros", "bilinear", sub_const=mn))
        if lab is None:
            b_label.append(None)
        else:
            b_label.append(ops.affine_sample(lab, theta, patch_size, "zeros", "nearest").long())
    return b_img, b_label


def get_argmaxed_segs(segs):
    with_bg = torch.cat([(segs.sum(1, keepdim=True) < 1.0).float(), segs], dim=1)
    return with_bg.argmax(1, keepdim=True)


def get_imgs(tta_sample):
    return tta_sample[:, 0:1]


def soft_dice_loss(smp_a, smp_b):
    """Reference: torch_utils.py:90-104 — per-class soft Dice [B,C] of two probability maps [B,C,D,H,W] (no eps; all
    ones when the denominators sum to zero).  Runs on the HIP reduction kernels (csrc/softdice.hip), differentiable.
    The TTA loop itself uses ops.consistency_loss, which fuses tta.py:263-269 (mask + softmax + this Dice + backward)."""
    return ops.soft_dice(smp_a, smp_b)


def dice_coeff(outputs, labels, max_label):
    """Reference: torch_utils.py:107-117. outputs/labels: integer maps of equal shape (on the GPU)."""
    b = outputs.shape[0]
    o = outputs.reshape(b, 1, *outputs.shape[-3:]).long()
    _, counts = ops.argmax_dice_from_labels(o, labels, max_label)
    n = float(outputs.numel())
    c = counts.cpu().to(torch.float32)
    pred, gt, both = c[0, 1:], c[1, 1:], c[2, 1:]
    return (2.0 * (both / n)) / (1e-8 + pred / n + gt / n)


def fix_all(m):
    for p in m.parameters():
        p.requires_grad_(False)


def release_all(m):
    for p in m.parameters():
        p.requires_grad_(True)


def release_norms(m):
    name = m.__class__.__name__.lower()
    if "instancenorm" in name or "batchnorm" in name:
        print("Released", m.__class__.__name__)
        for p in m.parameters():
            p.requires_grad_(True)


def register_forward_pre_hook_at_beginning(model, hook_fn):
    hooks = [hook_fn] + list(model._forward_pre_hooks.values())
    model._forward_pre_hooks = OrderedDict(zip(range(len(hooks)), hooks))


def register_forward_hook_at_beginning(model, hook_fn):
    hooks = [hook_fn] + list(model._forward_hooks.values())
    model._forward_hooks = OrderedDict(zip(range(len(hooks)), hooks))


def hookify(fn, type):
    assert type in ["forward_pre_hook", "forward_hook"]
    if type == "forward_pre_hook":
        hook = lambda module, input: fn(*input)
    else:
        hook = lambda module, input, output: fn(output)
    hook._dgtta_fn = fn          # (lets the engine see WHICH function a hook wraps: inference._mind_ahead_ok)
    return hook


def map_label(label, map_idxs, input_format):
    assert input_format in ["logits", "argmaxed"]
    if input_format == "logits":
        return label.transpose(0, 1)[map_idxs.to(label.device)].transpose(0, 1)
    mapped = torch.zeros_like(label)
    for new, old in enumerate(map_idxs):
        mapped[label == old] = new
    return mapped


def generate_label_mapping(source_label_dict, target_label_dict):
    assert all(isinstance(k, str) for k in source_label_dict.keys())
    assert all(isinstance(k, str) for k in target_label_dict.keys())
    assert set(source_label_dict.keys()).intersection(target_label_dict.keys()), \
        "There are no intersecting label names in given dicts."
    mapping = dict.fromkeys(list(source_label_dict.keys()) + list(target_label_dict.keys()))
    for key in mapping:
        if key in source_label_dict and key in target_label_dict:
            mapping[key] = (source_label_dict[key], target_label_dict[key])
    return {k: v for k, v in mapping.items() if v is not None}


def get_map_idxs(label_mapping: dict, optimized_labels: list, input_type):
    assert input_type in ["pretrain_labels", "tta_labels"]
    assert optimized_labels[0] == "background"
    col = 0 if input_type == "pretrain_labels" else 1
    return torch.as_tensor([label_mapping[lbl][col] for lbl in optimized_labels])
