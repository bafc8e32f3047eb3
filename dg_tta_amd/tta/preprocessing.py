"""Raw case -> network input: the job of nnU-Net's DefaultPreprocessor.run_case, which the reference calls from
preprocess_fromfile (dg_tta/tta/nnunet_utils.py:170-204) [3P nnunetv2==2.2.1: default_preprocessor.py, cropping.py,
default_normalization_schemes.py, default_resampling.py]: transpose_forward, crop to the non-zero bounding box (holes
filled; outside voxels of the label map become -1), per-channel normalisation (CTNormalization with the plans'
foreground intensity statistics, ZScoreNormalization), resampling to the configuration's spacing with the plans'
orders (image cubic spline, labels linear per label with a 0.5 threshold, nearest along a strongly anisotropic axis).

The resampling - the only part with real arithmetic volume - runs on the GPU (csrc/resample.hip: separable spline passes
with scipy.ndimage.zoom / skimage.resize semantics, float64 like the reference); cropping and normalisation are a few
numpy passes on the host, as in nnU-Net.  nnunetv2 is not installed here: restated from its published behaviour, parity
checked against oracle/preprocessing.py (scipy) only.  Reading: NIfTI-1 via nifti_io (SimpleITK order [z,y,x])."""
import numpy as np
import torch

from .. import ops
from .image_io import read_image as read_nifti      # NIfTI / NRRD / MetaImage, same (array [z,y,x], header) contract


def _resize(vol, new_shape, order, device, clip=True, axes=None):
    """skimage.transform.resize(vol, new_shape, order, mode='edge', anti_aliasing=False, clip=clip) on the GPU.
    vol: numpy [..., X, Y, Z]; returns a float64 numpy array."""
    t = torch.from_numpy(np.ascontiguousarray(vol, dtype=np.float64)).to(device)
    out = ops.resize_volume(t, new_shape, order, axes=axes)
    if clip and order > 1:          # (linear interpolation cannot leave the input range)
        # skimage clips to the range of the array it was called on: the whole channel volume, or - on nnU-Net's
        # separate-z path, where resize runs per 2-D slice - each slice's own range
        lead = t.dim() - 3
        red = tuple(range(lead, t.dim())) if axes is None else tuple(lead + a for a in axes)
        lo, hi = t.amin(dim=red, keepdim=True), t.amax(dim=red, keepdim=True)
        out = torch.maximum(torch.minimum(out, hi), lo)
    return out.cpu().numpy()


def create_nonzero_mask(data):
    from scipy.ndimage import binary_fill_holes
    mask = np.zeros(data.shape[1:], dtype=bool)
    for c in range(data.shape[0]):
        mask |= data[c] != 0
    return binary_fill_holes(mask)


def crop_to_nonzero(data, seg, nonzero_label=-1):
    mask = create_nonzero_mask(data)
    idx = np.where(mask)
    bbox = [[int(np.min(i)), int(np.max(i)) + 1] for i in idx]
    sl = tuple(slice(a, b) for a, b in bbox)
    data = data[(slice(None),) + sl]
    mask = mask[sl][None]
    if seg is not None:
        seg = seg[(slice(None),) + sl].copy()
        seg[(seg == 0) & (~mask)] = nonzero_label
    else:
        seg = np.where(mask, 0, nonzero_label).astype(np.int8)
    return data, seg, bbox


def normalize(data, seg, schemes, use_mask, props):
    out = data.astype(np.float32)
    for c in range(data.shape[0]):
        img = out[c]
        if schemes[c] == "CTNormalization":
            p = props[str(c)]
            img = np.clip(img, p["percentile_00_5"], p["percentile_99_5"])
            img = (img - p["mean"]) / max(p["std"], 1e-8)
        elif schemes[c] == "ZScoreNormalization":
            if use_mask[c]:
                m = seg[0] >= 0
                img = img.copy()
                img[m] = (img[m] - img[m].mean()) / max(img[m].std(), 1e-8)
            else:
                img = (img - img.mean()) / max(img.std(), 1e-8)
        elif schemes[c] != "NoNormalization":
            raise NotImplementedError(f"normalization scheme {schemes[c]}")
        out[c] = img
    return out


def compute_new_shape(old_shape, old_spacing, new_spacing):
    return [int(round(i / j * k)) for i, j, k in zip(old_spacing, new_spacing, old_shape)]


def separate_z(current_spacing, new_spacing, threshold=3):
    def aniso(sp):
        return (np.max(sp) / np.min(sp)) > threshold

    def lowres(sp):
        return np.where(max(sp) / np.array(sp) == 1)[0]
    axis = None
    if aniso(current_spacing):
        axis = lowres(current_spacing)
    elif aniso(new_spacing):
        axis = lowres(new_spacing)
    if axis is None or len(axis) != 1:
        return False, None
    return True, int(axis[0])


def resample_data_or_seg(data, new_shape, is_seg, axis, order, do_separate_z, order_z, device):
    """default_resampling.resample_data_or_seg: data [C,X,Y,Z] numpy -> [C,*new_shape] (same dtype)."""
    dtype = data.dtype
    shape = list(data[0].shape)
    new_shape = [int(s) for s in new_shape]
    if shape == new_shape:
        return data
    if order_z != 0:
        raise NotImplementedError("order_z != 0 (the plans use 0: nearest along the low-resolution axis)")
    inplane = [a for a in range(3) if a != axis] if do_separate_z else None        # separable: slices = per-axis passes
    mid_shape = [new_shape[a] if (not do_separate_z or a != axis) else shape[a] for a in range(3)]

    def finish(vol):      # nearest resampling along the anisotropic axis (map_coordinates order 0, mode nearest)
        if do_separate_z and shape[axis] != new_shape[axis]:
            t = torch.from_numpy(np.ascontiguousarray(vol)).to(device)
            vol = ops.resize_volume(t, new_shape, 0, axes=[axis]).cpu().numpy()
        return vol
    if not is_seg:
        out = finish(_resize(data, mid_shape, order, device, clip=True, axes=inplane))
        return out.astype(dtype)
    if order == 0:
        return finish(_resize(data.astype(float), mid_shape, 0, device, axes=inplane)).astype(dtype)
    # resize_segmentation: per label, linear resize of the indicator, threshold 0.5, labels written in ascending order.
    # With a separate low-resolution axis nnU-Net thresholds slice by slice first and then takes nearest slices.
    out = np.zeros((data.shape[0], *mid_shape), dtype=dtype)
    for c in range(data.shape[0]):
        labels = np.unique(data[c])
        stack = np.stack([(data[c] == l) for l in labels]).astype(np.float64)
        r = _resize(stack, mid_shape, order, device, clip=True, axes=inplane)
        for k, l in enumerate(labels):
            out[c][r[k] >= 0.5] = l
    return finish(out.astype(np.float64)).astype(dtype)


def run_case_npy(data, seg, spacing, plans, configuration, device="cuda"):
    """DefaultPreprocessor.run_case_npy: data [C,z,y,x], seg [1,z,y,x] or None, spacing (z,y,x)."""
    conf = plans["configurations"][configuration]
    while "inherits_from" in conf:
        parent = dict(plans["configurations"][conf["inherits_from"]])
        parent.update({k: v for k, v in conf.items() if k != "inherits_from"})
        conf = parent
    tf = plans["transpose_forward"]
    data = data.transpose([0] + [i + 1 for i in tf]).astype(np.float32)
    if seg is not None:
        seg = seg.transpose([0] + [i + 1 for i in tf])
    spacing = [float(spacing[i]) for i in tf]
    props = {"spacing": spacing, "shape_before_cropping": tuple(data.shape[1:])}
    data, seg, bbox = crop_to_nonzero(data, seg)
    props["bbox_used_for_cropping"] = bbox
    props["shape_after_cropping_and_before_resampling"] = tuple(data.shape[1:])
    target = list(conf["spacing"])
    if len(target) < 3:
        target = [spacing[0]] + target
    new_shape = compute_new_shape(data.shape[1:], spacing, target)
    data = normalize(data, seg, conf["normalization_schemes"], conf["use_mask_for_norm"],
                     plans["foreground_intensity_properties_per_channel"])
    do_sep, axis = separate_z(spacing, target)
    kd = conf.get("resampling_fn_data_kwargs", {"order": 3, "order_z": 0})
    ks = conf.get("resampling_fn_seg_kwargs", {"order": 1, "order_z": 0})
    data = resample_data_or_seg(data, new_shape, False, axis, kd["order"], do_sep, kd["order_z"], device)
    seg = resample_data_or_seg(seg, new_shape, True, axis, ks["order"], do_sep, ks["order_z"], device)
    seg = seg.astype(np.int16 if np.max(seg) > 127 else np.int8)
    return data, seg, props


def run_case(image_files, seg_file, plans, configuration, device="cuda"):
    """DefaultPreprocessor.run_case on NIfTI / NRRD / MetaImage files: one file per input channel, optional label file."""
    imgs, hdr = [], None
    for f in image_files:
        arr, hdr = read_nifti(f)
        imgs.append(arr.astype(np.float32))
    data = np.stack(imgs)
    spacing = tuple(hdr["pixdim"][::-1])                  # SimpleITK: GetSpacing()[::-1] -> (z, y, x)
    seg = None
    if seg_file is not None:
        seg = read_nifti(seg_file)[0][None]
    data, seg, props = run_case_npy(data, seg, spacing, plans, configuration, device)
    props["nifti_header"] = hdr
    return data, seg, props
