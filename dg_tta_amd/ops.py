"""Tensor-level wrappers over the C ABI (include/dgtta.h).  PyTorch is used for device memory, streams and autograd
glue only; every number is produced by the HIP kernels in dg_tta_amd/csrc.  All functions require CUDA(HIP) tensors.

Channel-last convention: a logical [B,C,D,H,W] tensor whose memory is [B,D,H,W,C] is exactly PyTorch's
`torch.channels_last_3d`; kernels that work voxel-major return such tensors, so callers keep NCDHW semantics.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, require_cuda, stream_of

F32, BF16, F16 = 0, 1, 2


def dtype_code(torch_dtype):
    """DGTTA_F32 / DGTTA_BF16 / DGTTA_F16 of a torch storage dtype."""
    try:
        return {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[torch_dtype]
    except KeyError:
        raise ValueError(f"unsupported activation dtype {torch_dtype}") from None
PAD_ZEROS, PAD_BORDER = 0, 1
LINEAR, NEAREST = 0, 1


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _f32c(t):
    return t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t


def is_cl3d(t):
    """True when a 5-D tensor's memory is dense [B,D,H,W,C]."""
    return t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d)


def empty_cl3d(b, c, d, h, w, dtype, device):
    return torch.empty((b, d, h, w, c), dtype=dtype, device=device).permute(0, 4, 1, 2, 3)


# ------------------------------------------------------------------------------------------------ MIND
def gauss_taps(sigma):
    """The smoothing kernel exactly as the reference evaluates it (mind.py:30-37, fp32 on the CPU)."""
    s = torch.tensor([float(sigma)])
    n = int(torch.ceil(s * 3.0 / 2.0).long().item()) * 2 + 1
    w = torch.exp(-torch.pow(torch.linspace(-(n // 2), n // 2, n), 2) / (2 * torch.pow(s, 2)))
    return (w / w.sum()).tolist()


def mind3d(img, noise, randn_weighting=0.05, out_format="ncdhw", out_ldc=12, out_dtype=torch.float32, groups=1, delta=1,
           sigma=1.0):
    """MIND3D descriptor of img [B,1,D,H,W] with the randn draw `noise` [B,12,D,H,W] (reference: dg_tta/mind.py:142-164).

    out_format 'ncdhw' -> contiguous [B,12,D,H,W] fp32; 'ndhwc' -> raw [B,D,H,W,out_ldc] buffer (fp32 or bf16).
    groups > 1: the batch is `groups` independent calls of B/groups samples (the variance clamp of mind.py:159-161
    uses the mean over the whole call's batch), written into one output buffer.
    """
    require_cuda(img, noise)
    lib = _lib.load()
    b, c, d, h, w = img.shape
    assert c == 1 and tuple(noise.shape) == (b, 12, d, h, w)
    img, noise = _f32c(img), _f32c(noise)
    ndhwc = out_format == "ndhwc"
    if ndhwc:
        out = torch.empty((b, d, h, w, out_ldc), dtype=out_dtype, device=img.device)
    else:
        out = torch.empty((b, 12, d, h, w), dtype=torch.float32, device=img.device)
    assert b % groups == 0
    bg = b // groups
    nbytes = lib.dgtta_mind3d_ws_bytes(bg, d, h, w)
    ws = _ws(nbytes, img.device)
    taps = gauss_taps(sigma)
    h_taps = (C.c_float * len(taps))(*taps)
    for g in range(groups):
        sl = slice(g * bg, (g + 1) * bg)
        check(lib.dgtta_mind3d_fwd(ptr(img[sl]), ptr(noise[sl]), float(randn_weighting), int(delta), h_taps, len(taps),
                                   ptr(out[sl]), int(ndhwc),
                                   int(out_ldc), dtype_code(out_dtype), ptr(ws), nbytes, bg, d, h, w,
                                   stream_of(img.device)), "dgtta_mind3d_fwd")
    return out


# ------------------------------------------------------------------------------------------------ GIN
def gin_chain(x, alpha, ks, kers, shifts):
    """GIN chain on x [B,1,D,H,W] with explicit draws (reference: dg_tta/gin.py:168-230)."""
    require_cuda(x, alpha, *kers, *shifts)
    lib = _lib.load()
    b, c, d, h, w = x.shape
    assert c == 1 and len(ks) == len(kers) == len(shifts) == 4
    x = _f32c(x)
    alpha = _f32c(alpha)
    kers = [_f32c(k) for k in kers]
    shifts = [_f32c(s) for s in shifts]
    out = torch.empty_like(x)
    nbytes = lib.dgtta_gin_ws_bytes(b, d, h, w)
    ws = _ws(nbytes, x.device)
    ksz = (C.c_int * 4)(*[int(k) for k in ks])
    kp = (C.c_void_p * 4)(*[k.data_ptr() for k in kers])
    sp = (C.c_void_p * 4)(*[s.data_ptr() for s in shifts])
    check(lib.dgtta_gin_chain_fwd(ptr(x), ptr(alpha), ksz, kp, sp, ptr(out), ptr(ws), nbytes, b, d, h, w,
                                  stream_of(x.device)), "dgtta_gin_chain_fwd")
    return out


# ------------------------------------------------------------------------------------------------ warp
def _warp_layout(t):
    """(tensor to pass, ndhwc flag, ldc)."""
    if is_cl3d(t) and t.shape[1] > 1:
        return t, 1, t.shape[1]
    t = t.contiguous()
    return t, 0, 0


def _warp_fwd_raw(src, theta, out_size, pad_mode, interp, algebra, sub_const):
    lib = _lib.load()
    b, c, ds, hs, wsz = src.shape
    dd, hd, wd = out_size
    src, ndhwc, ldc = _warp_layout(src)
    dst = empty_cl3d(b, c, dd, hd, wd, torch.float32, src.device) if ndhwc else \
        torch.empty((b, c, dd, hd, wd), dtype=torch.float32, device=src.device)
    check(lib.dgtta_affine_warp3d_fwd(ptr(src), ptr(theta), ptr(dst), b, c, ds, hs, wsz, dd, hd, wd, ndhwc, ldc, ldc,
                                      pad_mode, interp, int(algebra), ptr(sub_const), stream_of(src.device)),
          "dgtta_affine_warp3d_fwd")
    return dst


class _AffineWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, theta, out_size, pad_mode, algebra):
        ctx.save_for_backward(theta)
        ctx.meta = (tuple(src.shape), tuple(out_size), pad_mode, algebra, is_cl3d(src) and src.shape[1] > 1)
        return _warp_fwd_raw(src, theta, out_size, pad_mode, LINEAR, algebra, None)

    @staticmethod
    def backward(ctx, grad):
        (theta,) = ctx.saved_tensors
        shape, out_size, pad_mode, algebra, ndhwc = ctx.meta
        lib = _lib.load()
        b, c, ds, hs, wsz = shape
        dd, hd, wd = out_size
        if ndhwc:
            grad = grad.contiguous(memory_format=torch.channels_last_3d)
            gsrc = torch.empty((b, ds, hs, wsz, c), dtype=torch.float32, device=grad.device).permute(0, 4, 1, 2, 3)
        else:
            grad = grad.contiguous()
            gsrc = torch.empty(shape, dtype=torch.float32, device=grad.device)
        check(lib.dgtta_affine_warp3d_bwd(ptr(grad), ptr(theta), ptr(gsrc), b, c, ds, hs, wsz, dd, hd, wd, int(ndhwc), c,
                                          c, pad_mode, int(algebra), stream_of(grad.device)), "dgtta_affine_warp3d_bwd")
        return gsrc, None, None, None, None


def affine_warp(src, theta, out_size=None, padding_mode="zeros", tta_grid_algebra=False):
    """F.grid_sample(src, F.affine_grid(theta, ...), align_corners=False) (reference call sites tta.py:523-551,572-575).

    src [B,C,D,H,W] fp32 (contiguous or channels_last_3d; the layout is preserved), theta [B,3,4].
    Differentiable w.r.t. src (theta is a constant, as in the reference)."""
    require_cuda(src, theta)
    theta = _f32c(theta)
    src = src if src.dtype == torch.float32 else src.float()
    out_size = tuple(src.shape[2:]) if out_size is None else tuple(out_size)
    pad = PAD_BORDER if padding_mode == "border" else PAD_ZEROS
    return _AffineWarp.apply(src, theta, out_size, pad, bool(tta_grid_algebra))


def affine_sample(src, theta, out_size, padding_mode="zeros", mode="bilinear", sub_const=None):
    """Non-differentiable sampler with nearest mode and the (x - c) ... + c trick of get_batch (torch_utils.py:58-62)."""
    require_cuda(src, theta)
    pad = PAD_BORDER if padding_mode == "border" else PAD_ZEROS
    interp = NEAREST if mode == "nearest" else LINEAR
    return _warp_fwd_raw(src.float(), _f32c(theta), tuple(out_size), pad, interp, False, sub_const)


# ------------------------------------------------------------------------------------------------ loss
class _ConsistencyLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, la, lb, start_class, guard_items=0):
        lib = _lib.load()
        b, c = la.shape[:2]
        v = la.shape[2] * la.shape[3] * la.shape[4]
        la = la.contiguous(memory_format=torch.channels_last_3d)
        lb = lb.contiguous(memory_format=torch.channels_last_3d)
        dice = torch.empty((b, c), dtype=torch.float32, device=la.device)
        loss = torch.empty((), dtype=torch.float32, device=la.device)
        nbytes = lib.dgtta_softdice_ws_bytes(b, c, v)
        ws = _ws(nbytes, la.device)
        check(lib.dgtta_softdice_fwd(ptr(la), ptr(lb), ptr(dice), ptr(loss), ptr(ws), nbytes, b, c, v, c, start_class,
                                     guard_items if guard_items else b, stream_of(la.device)), "dgtta_softdice_fwd")
        ctx.save_for_backward(la, lb, ws)
        ctx.meta = (b, c, v, start_class)
        ctx.mark_non_differentiable(dice)
        return loss, dice

    @staticmethod
    def backward(ctx, gloss, _gdice):
        la, lb, ws = ctx.saved_tensors
        b, c, v, start_class = ctx.meta
        lib = _lib.load()
        ga = torch.empty_like(la, memory_format=torch.preserve_format)
        gb = torch.empty_like(lb, memory_format=torch.preserve_format)
        # the upstream gradient (e.g. 1/patches_to_be_accumulated) is read on the device: no host sync, no torch math
        gs = gloss.reshape(1).float().contiguous()
        check(lib.dgtta_softdice_bwd(ptr(la), ptr(lb), ptr(ga), ptr(gb), ptr(ws), 1.0, ptr(gs), b, c, v, c, start_class,
                                     stream_of(la.device)), "dgtta_softdice_bwd")
        return ga, gb, None, None


class _ConsistencyLossPair(torch.autograd.Function):
    """Same loss on the two halves of ONE batched tensor [2B,C,D,H,W] (branch a = first half): the gradient comes back
    as one buffer, so autograd needs no slice-backward zero fills and adds."""

    @staticmethod
    def forward(ctx, both, start_class, guard_items):
        lib = _lib.load()
        # the producer's offer to take the gradient in its 16-bit storage type (unet.Grad16Sink; 16-class rows only)
        sink = getattr(both, "_dgtta_grad16", None)
        ctx.sink = sink if (sink is not None and both.shape[1] == 16 and sink.claim()) else None
        both = both.contiguous(memory_format=torch.channels_last_3d)
        b2, c = both.shape[:2]
        b = b2 // 2
        v = both.shape[2] * both.shape[3] * both.shape[4]
        la, lb = both[:b], both[b:]
        dice = torch.empty((b, c), dtype=torch.float32, device=both.device)
        loss = torch.empty((), dtype=torch.float32, device=both.device)
        nbytes = lib.dgtta_softdice_ws_bytes(b, c, v)
        ws = _ws(nbytes, both.device)
        check(lib.dgtta_softdice_fwd(ptr(la), ptr(lb), ptr(dice), ptr(loss), ptr(ws), nbytes, b, c, v, c, start_class,
                                     guard_items if guard_items else b, stream_of(both.device)), "dgtta_softdice_fwd")
        ctx.save_for_backward(both, ws)
        ctx.meta = (b, c, v, start_class)
        ctx.mark_non_differentiable(dice)
        return loss, dice

    @staticmethod
    def backward(ctx, gloss, _gdice):
        both, ws = ctx.saved_tensors
        b, c, v, start_class = ctx.meta
        lib = _lib.load()
        gs = gloss.reshape(1).float().contiguous()
        if ctx.sink is not None:
            # the gradient goes to the producer in ITS storage type through the sink; autograd gets a stride-0 placeholder
            g16 = torch.empty((2 * b, *both.shape[2:], c), dtype=ctx.sink.dtype, device=both.device)
            check(lib.dgtta_softdice_bwd_t(ptr(both[:b]), ptr(both[b:]), ptr(g16[:b]), ptr(g16[b:]), ptr(ws), 1.0, ptr(gs), b, c, v,
                                           c, start_class, dtype_code(ctx.sink.dtype), stream_of(both.device)), "dgtta_softdice_bwd_t")
            ctx.sink.put(g16)
            return torch.zeros((), dtype=both.dtype, device=both.device).expand(both.shape), None, None
        g = torch.empty_like(both, memory_format=torch.preserve_format)
        check(lib.dgtta_softdice_bwd(ptr(both[:b]), ptr(both[b:]), ptr(g[:b]), ptr(g[b:]), ptr(ws), 1.0, ptr(gs), b, c, v, c,
                                     start_class, stream_of(both.device)), "dgtta_softdice_bwd")
        return g, None, None


def consistency_loss(target_a, target_b, start_class=1):
    """tta.py:263-269: masked softmax of both branches + `1 - soft_dice[:, start_class:].mean()`. Returns (loss, dice[B,C]).
    Targets produced by tta.calc_both_branches carry their common batched tensor (`_dgtta_pair`): the loss is then taken
    on that tensor directly."""
    require_cuda(target_a, target_b)
    pair = getattr(target_a, "_dgtta_pair", None)
    if pair is not None and pair is getattr(target_b, "_dgtta_pair", None) and pair.dtype == torch.float32:
        return _ConsistencyLossPair.apply(pair, int(start_class), int(getattr(target_a, "_dgtta_guard_items", 0)))
    return _ConsistencyLoss.apply(target_a.float(), target_b.float(), int(start_class),
                                  int(getattr(target_a, "_dgtta_guard_items", 0)))


class _DiceCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, smooth, do_bg):
        lib = _lib.load()
        b, c = logits.shape[:2]
        v = logits[0, 0].numel()
        logits = logits.contiguous(memory_format=torch.channels_last_3d)
        labels = labels.reshape(b, v).contiguous()
        loss3 = torch.empty(3, dtype=torch.float32, device=logits.device)
        dice = torch.empty((b, c), dtype=torch.float32, device=logits.device)
        nbytes = lib.dgtta_dice_ce_ws_bytes(b, c, v)
        ws = _ws(nbytes, logits.device)
        check(lib.dgtta_dice_ce_fwd(ptr(logits), c, ptr(labels), ptr(loss3), ptr(dice), ptr(ws), nbytes, b, c, v, float(smooth),
                                    int(do_bg), stream_of(logits.device)), "dgtta_dice_ce_fwd")
        ctx.save_for_backward(logits, labels, ws)
        ctx.meta = (b, c, v)
        ctx.mark_non_differentiable(dice)
        return loss3[0], dice, loss3[1:].detach()

    @staticmethod
    def backward(ctx, gloss, _gdice, _gparts):
        logits, labels, ws = ctx.saved_tensors
        b, c, v = ctx.meta
        lib = _lib.load()
        g = torch.empty_like(logits, memory_format=torch.preserve_format)
        gs = gloss.reshape(1).float().contiguous()
        check(lib.dgtta_dice_ce_bwd(ptr(logits), c, ptr(labels), ptr(ws), 1.0, ptr(gs), ptr(g), c, b, c, v,
                                    stream_of(logits.device)), "dgtta_dice_ce_bwd")
        return g, None, None, None


def dice_ce_loss(logits, labels, smooth=1e-5, do_bg=False):
    """nnU-Net's pre-training loss DC_and_CE_loss [3P] (soft Dice per sample without background + cross-entropy) of logits
    [B,C,D,H,W] (fp32) against an integer label map [B,1,D,H,W] / [B,D,H,W] (labels outside [0, C) are ignored).
    Returns (loss, dice [B,C], (ce, -mean dice)); differentiable w.r.t. the logits."""
    require_cuda(logits, labels)
    if labels.dtype != torch.int64:
        labels = labels.long()
    return _DiceCE.apply(logits.float(), labels, smooth, do_bg)


class _SoftDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        bsz, c = a.shape[:2]
        v = a[0, 0].numel()
        dense = [t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last_3d) for t in (a, b)]
        if not (all(dense) and a.stride() == b.stride()):
            a, b = a.contiguous(), b.contiguous()
        cl = a.dim() == 5 and not a.is_contiguous()             # dense channels-last
        strides = (v * c, 1, c) if cl else (v * c, v, 1)
        dice = torch.empty((bsz, c), dtype=torch.float32, device=a.device)
        nbytes = lib.dgtta_softdice_ws_bytes(bsz, c, v)
        ws = _ws(nbytes, a.device)
        check(lib.dgtta_softdice_probs_fwd(ptr(a), ptr(b), ptr(dice), ptr(ws), nbytes, bsz, c, v, *strides,
                                           stream_of(a.device)), "dgtta_softdice_probs_fwd")
        ctx.save_for_backward(a, b, ws)
        ctx.meta = (bsz, c, v, strides)
        return dice

    @staticmethod
    def backward(ctx, gdice):
        a, b, ws = ctx.saved_tensors
        bsz, c, v, strides = ctx.meta
        lib = _lib.load()
        ga, gb = torch.empty_like(a, memory_format=torch.preserve_format), torch.empty_like(b, memory_format=torch.preserve_format)
        check(lib.dgtta_softdice_probs_bwd(ptr(a), ptr(b), ptr(gdice.float().contiguous()), ptr(ga), ptr(gb), ptr(ws), bsz,
                                           c, v, *strides, stream_of(a.device)), "dgtta_softdice_probs_bwd")
        return ga, gb


def soft_dice(smp_a, smp_b):
    """soft_dice_loss(smp_a, smp_b) -> [B,C] of the reference (torch_utils.py:90-104) on probability maps [B,C,D,H,W]
    (fp32, contiguous or channels_last_3d); differentiable w.r.t. both inputs."""
    require_cuda(smp_a, smp_b)
    assert smp_a.shape == smp_b.shape and smp_a.dim() >= 3
    return _SoftDice.apply(smp_a.float(), smp_b.float())


# ------------------------------------------------------------------------------------------------ AdamW
def adamw_step(params, grads, exp_avgs, exp_avg_sqs, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
               grad_scale=1.0, skip_flag=None):
    """One multi-tensor AdamW step; skipped on the device when skip_flag (int32 device scalar) is non-zero."""
    lib = _lib.load()
    n = len(params)
    if n == 0:
        return
    require_cuda(*params)
    P = C.c_void_p * n
    hp = P(*[p.data_ptr() for p in params])
    hg = P(*[(g.data_ptr() if g is not None else None) for g in grads])
    hm = P(*[m.data_ptr() for m in exp_avgs])
    hv = P(*[v.data_ptr() for v in exp_avg_sqs])
    hn = (C.c_int64 * n)(*[p.numel() for p in params])
    check(lib.dgtta_adamw_step(hp, hg, hm, hv, hn, n, float(lr), float(betas[0]), float(betas[1]), float(eps),
                               float(weight_decay), int(step), float(grad_scale), ptr(skip_flag),
                               stream_of(params[0].device)), "dgtta_adamw_step")


def grads_nonfinite(grads, flag):
    """Sets flag (int32 device scalar, zeroed by the caller) when any element of `grads` is inf / NaN."""
    lib = _lib.load()
    grads = [g for g in grads if g is not None]
    if not grads:
        return
    require_cuda(*grads)
    n = len(grads)
    hg = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
    hn = (C.c_int64 * n)(*[g.numel() for g in grads])
    check(lib.dgtta_grads_nonfinite(hg, hn, n, ptr(flag), stream_of(grads[0].device)), "dgtta_grads_nonfinite")


# ------------------------------------------------------------------------------------------------ eval helpers
def argmax_dice(logits, labels=None):
    """argmax over channels (+ per-label counts for hard Dice, tta.py:321, torch_utils.py:107-117).

    Returns (argmax [B,D,H,W] int64, counts [3,C] int64 or None)."""
    require_cuda(logits)
    lib = _lib.load()
    b, c, d, h, w = logits.shape
    lg = logits.float().contiguous(memory_format=torch.channels_last_3d)
    am = torch.empty((b, d, h, w), dtype=torch.int64, device=lg.device)
    counts = None
    lp = None
    if labels is not None:
        labels = labels.reshape(b, d, h, w).to(torch.int64).contiguous()
        counts = torch.zeros((3, c), dtype=torch.int64, device=lg.device)
        lp = labels.data_ptr()
    check(lib.dgtta_argmax_dice(ptr(lg), c, c, lp, ptr(am), ptr(counts), b, d * h * w, stream_of(lg.device)),
          "dgtta_argmax_dice")
    return am, counts


def feature_head_argmax(facc, nsum, w, bsum):
    """Label map of feature-space window accumulators (csrc/window_features.hip): facc [M,X,Y,Z,32] fp32, nsum [X,Y,Z], w [M,C,32],
    bsum [C] = sum of the members' head biases -> int64 [X,Y,Z] = argmax_c sum_m (w[m,c] . facc[m,v]) + nsum[v] bsum[c]."""
    if facc.dim() == 4:
        facc, w = facc[None], (w[None] if w.dim() == 2 else w)
    if not (facc.is_cuda and facc.dtype == torch.float32 and facc[0].is_contiguous() and facc.shape[-1] == 32):
        raise ValueError("feature_head_argmax: fp32 GPU accumulators [M,X,Y,Z,32] expected")
    M, C = facc.shape[0], w.shape[1]
    if tuple(w.shape) != (M, C, 32) or tuple(bsum.shape) != (C,) or tuple(nsum.shape) != tuple(facc.shape[1:4]):
        raise ValueError("feature_head_argmax: w [M,C,32], bsum [C], nsum [X,Y,Z] expected")
    lib = _lib.load()
    w, bsum, nsum = w.float().contiguous(), bsum.float().contiguous(), nsum.float().contiguous()
    out = torch.empty(facc.shape[1:4], dtype=torch.int64, device=facc.device)
    check(lib.dgtta_feature_head_argmax(ptr(facc), facc.stride(0) if M > 1 else 0, ptr(nsum), ptr(w), ptr(bsum), M, 32, C, out.numel(),
                                        ptr(out), stream_of(facc.device)), "dgtta_feature_head_argmax")
    return out


def argmax_rows(acc):
    """Label map of a voxel-major accumulator [..., C] (fp32 or fp16, contiguous): argmax over the last axis, int64."""
    require_cuda(acc)
    if acc.dtype not in (torch.float32, torch.float16) or not acc.is_contiguous():
        raise ValueError("argmax_rows: contiguous fp32 / fp16 rows expected")
    lib = _lib.load()
    out = torch.empty(acc.shape[:-1], dtype=torch.int64, device=acc.device)
    check(lib.dgtta_argmax_rows(ptr(acc), F32 if acc.dtype == torch.float32 else F16, acc.shape[-1], out.numel(), ptr(out),
                                stream_of(acc.device)), "dgtta_argmax_rows")
    return out


def argmax_dice_from_labels(pred, labels, num_classes):
    """Per-label counts for two integer label maps (dice_coeff, torch_utils.py:107-117): counts [3,C] int64."""
    require_cuda(pred, labels)
    lib = _lib.load()
    pr = pred.reshape(-1).to(torch.int64).contiguous()
    lb = labels.reshape(-1).to(torch.int64).contiguous()
    assert pr.numel() == lb.numel()
    counts = torch.zeros((3, num_classes), dtype=torch.int64, device=pr.device)
    check(lib.dgtta_argmax_dice(None, 0, num_classes, ptr(lb), ptr(pr), ptr(counts), 1, pr.numel(),
                                stream_of(pr.device)), "dgtta_argmax_dice")
    return pr, counts


# ------------------------------------------------------------------------------------------------ resampling
def resize_volume(x, new_shape, order, axes=None):
    """skimage.transform.resize(x[c], new_shape, order, mode='edge', anti_aliasing=False, clip=False) for every leading
    index c, on the GPU: x [..., X, Y, Z] (any float dtype, CUDA) -> double [..., *new_shape].  `axes`: subset of the
    three trailing axes to resample (others must keep their size).  One separable pass per axis (csrc/resample.hip)."""
    require_cuda(x)
    lib = _lib.load()
    cur = x.double().contiguous()
    lead = cur.shape[:-3]
    outer0 = 1
    for s in lead:
        outer0 *= int(s)
    for ax in (range(3) if axes is None else axes):
        shp = list(cur.shape[-3:])
        n, m = shp[ax], int(new_shape[ax])
        if n == m:
            continue
        inner = 1
        for s in shp[ax + 1:]:
            inner *= s
        outer = outer0
        for s in shp[:ax]:
            outer *= s
        out_shape = shp[:ax] + [m] + shp[ax + 1:]
        dst = torch.empty((*lead, *out_shape), dtype=torch.float64, device=cur.device)
        nbytes = lib.dgtta_resample_axis_ws_bytes(outer, n, inner, int(order))
        ws = _ws(nbytes, cur.device)
        check(lib.dgtta_resample_axis(ptr(cur), ptr(dst), ptr(ws), nbytes, outer, n, m, inner, int(order),
                                      stream_of(cur.device)), "dgtta_resample_axis")
        cur = dst
    return cur
