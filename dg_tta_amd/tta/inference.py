"""Post-TTA ensemble sliding-window inference (SURVEY.md §8f "next" #1) — what the reference does at
dg_tta/tta/tta.py:379-416 through nnU-Net's predictor (dg_tta/tta/nnunet_utils.py:116-125, 208-230):
Gaussian-weighted 50 %-overlap sliding window, logits averaged over the ensemble's TTA'd parameter sets, argmax over
ALL pretrain classes, then map_label(argmaxed) to the target label ids.  Mirroring is off (the DG trainers set
inference_allowed_mirroring_axes=None, nnUNetTrainer_GIN_MIND.py:34-35).

The window logic is a restatement of nnunetv2==2.2.1 (`compute_gaussian`, `compute_steps_for_sliding_window`,
`predict_sliding_window_return_logits`) from its published behaviour: that package is not vendored with the reference, so
this stage is "parity unpinned" (checked against the CPU restatement in oracle/inference.py only).  Accumulators live
in HBM for the whole volume (105 classes x 512^3 = 56 GB in fp32 fits the 288 GB of an MI355X); they are fp32 by default
and fp16 - nnU-Net's storage type for `predicted_logits` - with DGTTA_WINDOW_ACC=fp16 or `acc_dtype=torch.float16` (sums
still formed in fp32, rounded once per window: half the read-modify-write traffic).  `export_segmentation` takes the logits back to the case's original geometry.

Round 5 - the FEATURE-space accumulator, what `predict_ensemble` / `run_inference` / `dgtta run_tta` use by default
(DGTTA_WINDOW_ACC=features): the segmentation head is linear and last, so sum_w g_w (W z_w + b) = W (sum_w g_w z_w) + b sum_w g_w;
the volume holds the 32 Gaussian-weighted feature channels the head reads (fp32: 16 GiB per member at 512^3 instead of 52.5 GiB of
logits, 2.8x less read-modify-write traffic per window) and the head runs once per voxel, fused with the argmax over the ensemble
(csrc/window_features.hip).  `predict_sliding_window_return_logits` / `predict_ensemble_logits` keep nnU-Net's logits-space form.
"""
import numpy as np
import torch

from .. import _lib, ops
from .._lib import check, ptr, stream_of
from ..mind import mind_groups
from .torch_utils import map_label


_GAUSS_CACHE = {}


def compute_gaussian(tile_size, sigma_scale=1.0 / 8, value_scaling_factor=10.0):
    """nnU-Net's importance map (cached per tile size: the scipy filter of a 128^3 tile costs ~0.15 s, more than the
    network passes of a small case)."""
    key = (tuple(int(t) for t in tile_size), float(sigma_scale), float(value_scaling_factor))
    if key not in _GAUSS_CACHE:
        _GAUSS_CACHE[key] = _compute_gaussian(key[0], sigma_scale, value_scaling_factor)
    return _GAUSS_CACHE[key].clone()


def _compute_gaussian(tile_size, sigma_scale, value_scaling_factor):
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(tile_size)
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode="constant", cval=0)
    g = torch.from_numpy(g).float()
    g = g / g.max() * value_scaling_factor
    g[g == 0] = g[g != 0].min()
    return g


def compute_steps_for_sliding_window(image_size, tile_size, tile_step_size=0.5):
    assert all(i >= j for i, j in zip(image_size, tile_size)), "image size must be as large or larger than patch_size"
    assert 0 < tile_step_size <= 1
    target = [i * tile_step_size for i in tile_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]
    steps = []
    for dim in range(len(tile_size)):
        max_step = image_size[dim] - tile_size[dim]
        actual = max_step / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def pad_to_patch(data, patch_size):
    """Centred zero padding up to the patch size (nnU-Net pads with constant 0); returns (padded, crop slices)."""
    pads, crop = [], []
    for have, want in zip(data.shape[1:], patch_size):
        extra = max(want - have, 0)
        lo = extra // 2
        pads.append((lo, extra - lo))
        crop.append(slice(lo, lo + have))
    if any(p != (0, 0) for p in pads):
        flat = [v for p in reversed(pads) for v in p]
        data = torch.nn.functional.pad(data, flat, mode="constant", value=0)
    return data, crop


import os as _os

# windows per network pass.  Measured at 512^3: 2 / 4 / 8 / 16 windows per pass = 3.26 / 3.14 / 3.07 / 3.03 ms per window in round 3;
# round 5 (feature-space accumulator): 0.541 s per member with 8, 0.537 with 12, 0.533 with 16 (~25 GB of activations in 16-bit storage)
WINDOW_BATCH = max(1, int(_os.environ.get("DGTTA_WINDOW_BATCH", "16")))


def window_acc_mode():
    """DGTTA_WINDOW_ACC = features (default: feature-space accumulator where the network offers it, else fp32 logits) | fp32 | fp16
    (logits-space accumulator of that storage type)."""
    v = _os.environ.get("DGTTA_WINDOW_ACC", "features").lower()
    if v not in ("features", "fp32", "fp16"):
        raise ValueError(f"DGTTA_WINDOW_ACC={v!r}: features, fp32 or fp16")
    return v


def window_acc_dtype():
    """Storage type of the LOGITS-space window accumulator: fp16 with DGTTA_WINDOW_ACC=fp16, else fp32."""
    return torch.float16 if window_acc_mode() == "fp16" else torch.float32


def _acc_code(acc):
    if acc.dtype not in (torch.float32, torch.float16):
        raise ValueError("the window accumulator is fp32 or fp16")
    return ops.F32 if acc.dtype == torch.float32 else ops.F16


def _inner(model):
    return getattr(model, "_orig_mod", model)


def _num_classes(model):
    return _inner(model).decoder.seg_layers[-1].out_channels


def _can_fuse_head_accumulate(model):
    """The head may write straight into the window accumulator when the network offers it and nothing stands between the
    head and the accumulation: every forward hook is the plan's untouched model-output modifier (identity)."""
    m = _inner(model)
    if not hasattr(m, "can_fuse_window_accumulate") or not m.can_fuse_window_accumulate():
        return False
    return _only_identity_output_hooks(m)


def _only_identity_output_hooks(m):
    from .config_log_utils import is_template_modifier
    for h in m._forward_hooks.values():
        cells = getattr(h, "__closure__", None) or ()
        fns = [c.cell_contents for c in cells if callable(c.cell_contents)]
        if len(fns) != 1 or not is_template_modifier(fns[0], "modfify_tta_model_output_fn"):
            return False
    return True


FEATURE_HEAD_MAX_CLASSES_MFMA = 128      # dgtta_feature_head_argmax: classes x 32 voxels per wave on the matrix cores
FEATURE_HEAD_MAX_CLASSES_VALU = 160      # its vector-ALU kernel with M > 1 members parks C * 256 * 4 B of partial sums in LDS


def _can_accumulate_features(model, members=1):
    """Feature-space accumulation needs the network's offer (head = 1x1x1 conv on 32 channels, nothing selected), nothing
    between the head and the accumulation (every forward hook is the plan's untouched model-output modifier = identity), and a
    head the final head + argmax kernel takes (C <= 128, or one member, or C <= 160) - checked UP FRONT, so that a
    wider ensemble goes to the logits-space accumulator instead of failing after every member's sliding-window pass."""
    m = _inner(model)
    if not (hasattr(m, "can_fuse_window_feature_accumulate") and m.can_fuse_window_feature_accumulate() and
            _only_identity_output_hooks(m)):
        return False
    ncls = _num_classes(m)
    return ncls <= FEATURE_HEAD_MAX_CLASSES_MFMA or members == 1 or ncls <= FEATURE_HEAD_MAX_CLASSES_VALU


class WindowFeatures:
    """Feature-space window accumulator of an ensemble: facc [M,X,Y,Z,32] (fp32, sum over windows of gauss * head input of member m),
    nsum [X,Y,Z] (sum of the window Gaussians, shared), crop (slices back from the padding), w [M,C,32] / b [M,C] the members' head
    weights.  The ensemble's accumulated logits - what predict_ensemble_logits returns as `acc` - are
    sum_m (w[m] @ facc[m] + b[m] * nsum); they are never formed for the whole volume."""

    def __init__(self, facc, nsum, crop, w, b):
        self.facc, self.nsum, self.crop, self.w, self.b = facc, nsum, crop, w.contiguous(), b.contiguous()
        self.bsum = b.sum(0).contiguous()

    @property
    def shape(self):
        return (*self.facc.shape[1:4], self.w.shape[1])

    def argmax(self):
        """Label map [X,Y,Z] (int64, padded geometry): argmax over all classes of the ensemble logits, first maximum wins."""
        return ops.feature_head_argmax(self.facc, self.nsum, self.w, self.bsum)

    def logits(self):
        """The accumulated ensemble logits [X,Y,Z,C] by torch (fp32) - for tests and small volumes only."""
        M, X, Y, Z, K = self.facc.shape
        out = (self.nsum[..., None] * self.bsum).float()
        for m in range(M):
            out = out + self.facc[m].reshape(-1, K).matmul(self.w[m].t()).reshape(X, Y, Z, -1)
        return out


def _mind_ahead_ok(model, dev):
    """The MIND descriptor of the NEXT window batch may be evaluated on a side stream while the network is in the current one
    (as tta.tta_epoch does for the next pass) when mind_hook is the last forward pre-hook and whatever stands in front of it
    leaves the input alone (gin_hook with the trainers' internal augmentation off, the plan's untouched input modifier).
    DGTTA_INFER_PREFETCH=0: in line."""
    import os
    from ..gin import gin_hook
    from ..mind import hook_owner, mind_hook
    from ..utils import get_internal_augmentation_enabled
    if torch.device(dev).type != "cuda" or os.environ.get("DGTTA_INFER_PREFETCH", "1") == "0":
        return False
    hooks = list(hook_owner(model)._forward_pre_hooks.values())
    if not hooks or hooks[-1] is not mind_hook or get_internal_augmentation_enabled():
        return False
    from .config_log_utils import is_template_modifier

    def leaves_input_alone(h):      # gin_hook (augmentation off, checked above) or the untouched template input modifier
        fn = getattr(h, "_dgtta_fn", None)
        return h is gin_hook or (fn is not None and is_template_modifier(fn, "modify_tta_input_fn"))
    return all(h is mind_hook or leaves_input_alone(h) for h in hooks) and sum(h is mind_hook for h in hooks) == 1


def _window_batches(model, data, origins, patch_size, dev):
    """Yields (group, work) per batch of WINDOW_BATCH windows.  Where _mind_ahead_ok, the window stack and its MIND descriptor
    (noise draw included: the draws keep their order, one batch after the other) are produced on a side stream one batch ahead
    and handed to the model's mind_hook (mind.push_features): same kernels, same draws, same result."""
    from .._state import state_of
    from ..mind import MIND3D, hook_owner, push_features

    def stack(group):
        return torch.stack([data[:, sx:sx + patch_size[0], sy:sy + patch_size[1], sz:sz + patch_size[2]]
                            for sx, sy, sz in group]).contiguous()

    groups = [origins[g0:g0 + WINDOW_BATCH] for g0 in range(0, len(origins), WINDOW_BATCH)]
    if not groups:
        return
    if not _mind_ahead_ok(model, dev):
        for group in groups:
            yield group, stack(group)
        return
    main = torch.cuda.current_stream(dev)
    side = state_of(hook_owner(model)).stream("prep_stream", dev)
    adt = getattr(_inner(model), "act_dtype", torch.float32)
    side.wait_stream(main)          # once: the volume (uploaded on the main stream) is visible to the side stream

    def prepare(group):
        with torch.cuda.stream(side):
            work = stack(group)
            feat = MIND3D().forward(work, out_dtype=adt, groups=len(group))
        for t in (work, feat):
            t.record_stream(main)
        return work, feat

    nxt = prepare(groups[0])
    try:
        for i, group in enumerate(groups):
            work, feat = nxt
            main.wait_stream(side)
            nxt = prepare(groups[i + 1]) if i + 1 < len(groups) else None      # runs beside this batch's network pass
            push_features(model, feat)
            yield group, work
    finally:
        state_of(hook_owner(model)).features.clear()      # (a descriptor no mind_hook call took must not meet a later input)


@torch.no_grad()
def predict_sliding_window_return_logits(model, data, patch_size, acc=None, tile_step_size=0.5, acc_dtype=None):
    """data [C,X,Y,Z] (CPU or GPU) -> accumulates gauss-weighted logits of `model` into acc [X,Y,Z,ncls] (GPU; fp32 or
    fp16: the dtype of a given `acc`, else `acc_dtype`, else window_acc_dtype()).
    Returns (acc, nsum, crop) ; logits = acc / nsum[..., None] cropped by `crop`."""
    if acc_dtype is None:
        acc_dtype = acc.dtype if acc is not None else window_acc_dtype()
    if acc_dtype not in (torch.float32, torch.float16):
        raise ValueError(f"the window accumulator is fp32 or fp16, not {acc_dtype}")
    if acc is not None and acc.dtype != acc_dtype:
        raise ValueError(f"acc is {acc.dtype} but acc_dtype = {acc_dtype}")
    if acc is not None and not (acc.is_cuda and acc.is_contiguous()):
        raise ValueError("acc must be a contiguous GPU tensor [X,Y,Z,classes]")
    lib = _lib.load()
    dev = next(model.parameters()).device
    data, crop = pad_to_patch(data.float(), patch_size)
    data = data.to(dev)
    X, Y, Z = data.shape[1:]
    gauss = compute_gaussian(tuple(patch_size)).to(dev).contiguous()
    steps = compute_steps_for_sliding_window((X, Y, Z), patch_size, tile_step_size)
    nsum = torch.zeros((X, Y, Z), dtype=torch.float32, device=dev)
    was_training = model.training
    model.eval()
    # windows go through the network WINDOW_BATCH at a time: the 16^3-and-below layers are launch / occupancy bound at
    # batch 1 (same observation as in the TTA loop); per-sample InstanceNorm and per-window MIND statistics (below) make
    # the result independent of the grouping
    origins = [(sx, sy, sz) for sx in steps[0] for sy in steps[1] for sz in steps[2]]
    for group, work in _window_batches(model, data, origins, patch_size, dev):
        # MIND's variance clamp uses the mean over the whole CALL's batch (mind.py:159-161) and nnU-Net predicts one
        # window per call: the batched pass keeps per-window statistics (groups = windows in the batch)
        if _can_fuse_head_accumulate(model):
            # the head evaluates straight into the accumulator (the windows' logits - 880 MB each at 128^3 x 105 - are not written)
            if acc is None:
                acc = torch.zeros((X, Y, Z, _num_classes(model)), dtype=acc_dtype, device=dev)
            with mind_groups(model, len(group)), _inner(model).fuse_window_accumulate(acc, nsum, gauss, group):
                model(work)
            continue
        with mind_groups(model, len(group)):
            out = model(work)
        if isinstance(out, tuple):
            out = out[0]
        out = out.float().contiguous(memory_format=torch.channels_last_3d)        # [n,C,P] stored voxel-major
        ncls = out.shape[1]
        if acc is None:
            acc = torch.zeros((X, Y, Z, ncls), dtype=acc_dtype, device=dev)
        for k, (sx, sy, sz) in enumerate(group):       # overlapping windows: accumulated one after the other
            check(lib.dgtta_window_accumulate_t(ptr(out[k]), ptr(gauss), ptr(acc), ptr(nsum), ncls, *patch_size, X, Y, Z, sx,
                                                sy, sz, _acc_code(acc), stream_of(dev)), "dgtta_window_accumulate_t")
    model.train(was_training)
    return acc, nsum, crop


@torch.no_grad()
def accumulate_window_features(model, data, patch_size, facc=None, tile_step_size=0.5):
    """The feature-space form of predict_sliding_window_return_logits for one member: data [C,X,Y,Z] -> adds gauss * (the head's
    input of every window) into facc [X,Y,Z,32] (fp32, GPU; allocated when None).  Returns (facc, nsum, crop)."""
    m = _inner(model)
    if not _can_accumulate_features(model):
        raise RuntimeError("accumulate_window_features: the network does not offer the feature-space accumulation here")
    dev = next(model.parameters()).device
    data, crop = pad_to_patch(data.float(), patch_size)
    data = data.to(dev)
    X, Y, Z = data.shape[1:]
    if facc is None:
        facc = torch.zeros((X, Y, Z, 32), dtype=torch.float32, device=dev)
    if tuple(facc.shape) != (X, Y, Z, 32) or facc.dtype != torch.float32 or not (facc.is_cuda and facc.is_contiguous()):
        raise ValueError(f"facc must be a contiguous fp32 GPU tensor [{X},{Y},{Z},32]")
    gauss = compute_gaussian(tuple(patch_size)).to(dev).contiguous()
    steps = compute_steps_for_sliding_window((X, Y, Z), patch_size, tile_step_size)
    nsum = torch.zeros((X, Y, Z), dtype=torch.float32, device=dev)
    was_training = model.training
    model.eval()
    origins = [(sx, sy, sz) for sx in steps[0] for sy in steps[1] for sz in steps[2]]
    for group, work in _window_batches(model, data, origins, patch_size, dev):
        with mind_groups(model, len(group)), m.fuse_window_feature_accumulate(facc, nsum, gauss, group):
            model(work)
    model.train(was_training)
    return facc, nsum, crop


@torch.no_grad()
def predict_ensemble_features(data, model, parameter_sets, patch_size):
    """Feature-space accumulators of the whole ensemble for one preprocessed case (see WindowFeatures); one [X,Y,Z,32] fp32
    volume per member, because the members' heads differ."""
    if hasattr(model, "set_selected_classes"):
        model.set_selected_classes(None)          # argmax runs over ALL pretrain classes, as in the reference
    dev = next(model.parameters()).device
    data = data.float().to(dev)      # one upload for all ensemble members
    padded, _ = pad_to_patch(data[:1], patch_size)
    facc = torch.zeros((len(parameter_sets), *padded.shape[1:], 32), dtype=torch.float32, device=dev)
    del padded
    ws, bs, nsum, crop = [], [], None, None
    for k, params in enumerate(parameter_sets):
        model.load_state_dict(params)
        head = _inner(model).decoder.seg_layers[-1]
        ws.append(head.weight.detach().reshape(head.out_channels, -1).float().clone())
        bs.append(head.bias.detach().float().clone())
        _, nsum, crop = accumulate_window_features(model, data, patch_size, facc[k])
    return WindowFeatures(facc, nsum, crop, torch.stack(ws), torch.stack(bs))


@torch.no_grad()
def predict_ensemble(data, model, parameter_sets, patch_size):
    """What run_inference / dgtta run_tta predict with: (acc, nsum, crop) for export_segmentation - acc a WindowFeatures when
    DGTTA_WINDOW_ACC=features (default) and the network offers it, else the logits-space accumulator of predict_ensemble_logits."""
    if window_acc_mode() == "features" and _can_accumulate_features(model, len(parameter_sets)):
        feats = predict_ensemble_features(data, model, parameter_sets, patch_size)
        return feats, feats.nsum, feats.crop
    return predict_ensemble_logits(data, model, parameter_sets, patch_size)


@torch.no_grad()
def predict_ensemble_logits(data, model, parameter_sets, patch_size):
    """Window-accumulated logits of the whole ensemble for one preprocessed case: data [C,X,Y,Z] (image channel(s) only),
    parameter_sets: list of state dicts (the `*_tta_parameters.pt` contents).  Returns (acc [X,Y,Z,ncls] fp32 on the GPU =
    sum over members and windows of gauss * logits, nsum [X,Y,Z], crop): the ensemble-mean logits nnU-Net's predictor hands
    on are acc / nsum / len(parameter_sets) inside `crop`."""
    if hasattr(model, "set_selected_classes"):
        model.set_selected_classes(None)          # argmax runs over ALL pretrain classes, as in the reference
    acc, nsum, crop = None, None, None
    data = data.float().to(next(model.parameters()).device)      # one upload for all ensemble members
    for params in parameter_sets:
        model.load_state_dict(params)
        acc, nsum, crop = predict_sliding_window_return_logits(model, data, patch_size, acc)
    return acc, nsum, crop


@torch.no_grad()
def run_inference(data, model, parameter_sets, patch_size, label_mapping=None, optimized_labels=None):
    """Ensemble prediction of one preprocessed case in the PREPROCESSED geometry.  Returns the label map [X,Y,Z] (int64,
    CPU), mapped to the target ids when label_mapping/optimized_labels are given (tta.py:407-411)."""
    from .torch_utils import get_map_idxs
    acc, nsum, crop = predict_ensemble(data, model, parameter_sets, patch_size)
    seg = torch.as_tensor(export_segmentation(acc, nsum, crop, None, None, None).astype(np.int64))
    if label_mapping is not None:
        seg = map_label(seg[None], get_map_idxs(label_mapping, optimized_labels, "pretrain_labels"), "argmaxed")[0]
    return seg


EXPORT_CLASS_GROUP = 8


def _resolved_configuration(plans, configuration):
    conf = plans["configurations"][configuration]
    while "inherits_from" in conf:
        parent = dict(plans["configurations"][conf["inherits_from"]])
        parent.update({k: v for k, v in conf.items() if k != "inherits_from"})
        conf = parent
    return conf


@torch.no_grad()
def export_segmentation(acc, nsum, crop, properties, plans, configuration):
    """Label map (pretrain class ids, numpy) of accumulated ensemble logits.

    properties None: argmax in the preprocessed geometry (sum over members of acc_m / nsum has the argmax of sum_m acc_m,
    nsum > 0 being shared: no division pass).  Otherwise nnU-Net's
    convert_predicted_logits_to_segmentation_with_correct_shape [3P nnunetv2==2.2.1, reached from
    dg_tta/tta/nnunet_utils.py:208-230]: the normalised logits are resampled to
    properties['shape_after_cropping_and_before_resampling'] with the plans' probability resampler (linear; nearest along
    a strongly anisotropic axis), argmax'd (softmax is monotone), pasted into a zero volume of 'shape_before_cropping' at
    'bbox_used_for_cropping' and transposed back with the plans' transpose_backward.  The resampling runs on the GPU class
    group by class group with a running argmax, so the full-resolution 105-class volume never exists."""
    from .preprocessing import separate_z
    lib = _lib.load()
    feats = acc if isinstance(acc, WindowFeatures) else None
    dev = nsum.device
    X, Y, Z, C = acc.shape
    st = stream_of(dev)
    label_map = (lambda: feats.argmax()) if feats is not None else (lambda: ops.argmax_rows(acc))
    cs = [(sl.start or 0, (sl.stop if sl.stop is not None else dim) - (sl.start or 0)) for sl, dim in zip(crop, (X, Y, Z))]
    (x0, xs), (y0, ys), (z0, zs) = cs
    cur_shape = [xs, ys, zs]
    if properties is None:
        return label_map()[tuple(crop)].cpu().numpy()
    tgt_shape = [int(v) for v in properties["shape_after_cropping_and_before_resampling"]]
    if tgt_shape == cur_shape:
        seg = label_map()[tuple(crop)].cpu().numpy()
    else:
        conf = _resolved_configuration(plans, configuration)
        cur_spacing = list(conf["spacing"])
        if len(cur_spacing) < 3:
            cur_spacing = [properties["spacing"][0]] + cur_spacing
        kw = conf.get("resampling_fn_probabilities_kwargs", {"order": 1, "order_z": 0})
        order, order_z = int(kw.get("order", 1)), int(kw.get("order_z", 0))
        if order not in (0, 1) or order_z != 0:
            raise NotImplementedError("probability resampling is built for order 0 / 1 and order_z 0 (nnU-Net's defaults)")
        do_sep, axis = separate_z(cur_spacing, properties["spacing"])
        vout = tgt_shape[0] * tgt_shape[1] * tgt_shape[2]
        best_val = torch.empty(vout, dtype=torch.float64, device=dev)
        best_idx = torch.empty(vout, dtype=torch.int32, device=dev)
        for c0 in range(0, C, EXPORT_CLASS_GROUP):
            cg = min(EXPORT_CLASS_GROUP, C - c0)
            cur = torch.empty((xs, ys, zs, cg), dtype=torch.float64, device=dev)
            if feats is not None:
                M = feats.facc.shape[0]
                check(lib.dgtta_feature_logits_chunk_f64(ptr(feats.facc), feats.facc.stride(0), ptr(nsum), ptr(feats.w), ptr(feats.bsum),
                                                         ptr(cur), M, 32, C, X, Y, Z, x0, y0, z0, xs, ys, zs, c0, cg, st),
                      "dgtta_feature_logits_chunk_f64")
            else:
                check(lib.dgtta_logits_chunk_f64_t(ptr(acc), ptr(nsum), ptr(cur), C, X, Y, Z, x0, y0, z0, xs, ys, zs, c0, cg,
                                                   _acc_code(acc), st), "dgtta_logits_chunk_f64_t")
            for ax in range(3):          # channels-last: the class group rides in `inner` of every pass
                n, m = cur.shape[ax], tgt_shape[ax]
                if n == m:
                    continue
                shp = list(cur.shape[:3])
                outer = int(np.prod(shp[:ax], dtype=np.int64))
                inner = int(np.prod(shp[ax + 1:], dtype=np.int64)) * cg
                dst = torch.empty((*shp[:ax], m, *shp[ax + 1:], cg), dtype=torch.float64, device=dev)
                o = 0 if (do_sep and ax == axis) else order      # nearest slices along the low-resolution axis
                check(lib.dgtta_resample_axis(ptr(cur), ptr(dst), None, 0, outer, n, m, inner, o, st), "dgtta_resample_axis")
                cur = dst
            check(lib.dgtta_argmax_merge_f64(ptr(cur), vout, cg, c0, ptr(best_val), ptr(best_idx), int(c0 == 0), st),
                  "dgtta_argmax_merge_f64")
        seg = best_idx.reshape(tgt_shape).cpu().numpy()
    full = np.zeros(tuple(int(v) for v in properties["shape_before_cropping"]),
                    dtype=np.uint8 if C - 1 < 255 else np.uint16)
    sl = tuple(slice(int(a), int(b)) for a, b in properties["bbox_used_for_cropping"])
    full[sl] = seg
    return np.ascontiguousarray(full.transpose(plans["transpose_backward"]))
