"""nnU-Net DefaultPreprocessor.run_case, CPU restatement (oracle; test infrastructure).

Third-party arithmetic: nnunetv2==2.2.1 (preprocessing/preprocessors/default_preprocessor.py, cropping/cropping.py,
normalization/default_normalization_schemes.py, resampling/default_resampling.py) reached from the reference's
preprocess_fromfile (/root/reference/dg_tta/tta/nnunet_utils.py:170-204); that package and its dependency scikit-image
are not installed here, so this follows their published algorithm: skimage.transform.resize(order, mode='edge',
anti_aliasing=False, clip=True) is scipy.ndimage.zoom(order, mode='nearest', grid_mode=True) + clipping to the input
range (scipy IS installed and is the numeric reference for the resampling).  parity unpinned by the reference.
"""
import numpy as np
from scipy import ndimage as ndi


def resize(image, new_shape, order, clip=True):
    image = np.asarray(image, dtype=np.float64)
    new_shape = tuple(int(s) for s in new_shape)
    if image.shape == new_shape:
        return image.copy()
    zoom = [o / i for o, i in zip(new_shape, image.shape)]
    out = ndi.zoom(image, zoom, order=order, mode="nearest", grid_mode=True)
    if clip and order != 0:
        out = np.clip(out, image.min(), image.max())
    return out


def resize_segmentation(seg, new_shape, order):
    if order == 0:
        return resize(seg.astype(float), new_shape, 0).astype(seg.dtype)
    out = np.zeros(new_shape, dtype=seg.dtype)
    for c in np.unique(seg):
        out[resize((seg == c).astype(float), new_shape, order) >= 0.5] = c
    return out


def create_nonzero_mask(data):
    mask = np.zeros(data.shape[1:], dtype=bool)
    for c in range(data.shape[0]):
        mask |= data[c] != 0
    return ndi.binary_fill_holes(mask)


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
            raise NotImplementedError(schemes[c])
        out[c] = img
    return out


def compute_new_shape(old_shape, old_spacing, new_spacing):
    return [int(round(i / j * k)) for i, j, k in zip(old_spacing, new_spacing, old_shape)]


def separate_z(current_spacing, new_spacing, threshold=3):
    """(do_separate_z, axis) of resample_data_or_seg_to_shape with force_separate_z=None."""
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


def resample_data_or_seg(data, new_shape, is_seg, axis, order, do_separate_z, order_z=0):
    dtype = data.dtype
    shape, new_shape = np.array(data[0].shape), np.array(new_shape)
    if not np.any(shape != new_shape):
        return data
    fn = (lambda a, s, o: resize_segmentation(a, s, o)) if is_seg else (lambda a, s, o: resize(a, s, o))
    data = data.astype(float)
    out = []
    for c in range(data.shape[0]):
        if do_separate_z:
            shape2d = tuple(int(s) for i, s in enumerate(new_shape) if i != axis)
            slices = [fn(np.take(data[c], k, axis=axis), shape2d, order) for k in range(shape[axis])]
            vol = np.stack(slices, axis)
            if shape[axis] != new_shape[axis]:
                grids = np.mgrid[tuple(slice(0, int(s)) for s in new_shape)].astype(float)
                for a in range(3):
                    grids[a] = float(vol.shape[a]) / new_shape[a] * (grids[a] + 0.5) - 0.5
                assert order_z == 0, "only order_z = 0 (the plans' value) is restated"
                vol = ndi.map_coordinates(vol, grids, order=0, mode="nearest")
            out.append(vol[None])
        else:
            out.append(fn(data[c], tuple(int(s) for s in new_shape), order)[None])
    return np.vstack(out).astype(dtype)


def run_case_npy(data, seg, spacing, plans, configuration):
    """data [C,z,y,x], seg [1,z,y,x] or None, spacing (z,y,x) -> (data float32, seg int8/int16, properties)."""
    conf = plans["configurations"][configuration]
    tf = plans["transpose_forward"]
    data = data.transpose([0] + [i + 1 for i in tf]).astype(np.float32)
    if seg is not None:
        seg = seg.transpose([0] + [i + 1 for i in tf])
    spacing = [spacing[i] for i in tf]
    props = {"spacing": spacing, "shape_before_cropping": data.shape[1:]}
    data, seg, bbox = crop_to_nonzero(data, seg)
    props["bbox_used_for_cropping"] = bbox
    props["shape_after_cropping_and_before_resampling"] = data.shape[1:]
    target = conf["spacing"]
    new_shape = compute_new_shape(data.shape[1:], spacing, target)
    data = normalize(data, seg, conf["normalization_schemes"], conf["use_mask_for_norm"],
                     plans["foreground_intensity_properties_per_channel"])
    do_sep, axis = separate_z(spacing, target)
    kd = conf.get("resampling_fn_data_kwargs", {"order": 3, "order_z": 0})       # nnU-Net's defaults
    ks = conf.get("resampling_fn_seg_kwargs", {"order": 1, "order_z": 0})
    data = resample_data_or_seg(data, new_shape, False, axis, kd["order"], do_sep, kd["order_z"])
    seg = resample_data_or_seg(seg, new_shape, True, axis, ks["order"], do_sep, ks["order_z"])
    seg = seg.astype(np.int16 if np.max(seg) > 127 else np.int8)
    return data, seg, props
