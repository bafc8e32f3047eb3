"""SimulateDiscreteLowResolutionTransform — drop-in for dg_tta/pretraining/discrete_downsampling.py:8-72 (the MultiRes
trainers' replacement for batchgenerators' SimulateLowResolutionTransform, nnUNetTrainer_GIN_MIND_MultiRes.py:54-66):
per sample (p_per_sample) draw a zoom factor per axis from the discrete `zoom_range`, per channel (p_per_channel)
downsample to round(shape * zoom) and upsample back with skimage.transform.resize(mode='edge', anti_aliasing=False).

The resizes run on the GPU (csrc/resample.hip: the separable spline passes that also serve the preprocessing; orders 0, 1
and 3 with scipy.ndimage.zoom / skimage semantics incl. the clip to the input range).  Random draws follow the reference's
order on numpy's global generator: uniform (sample), choice (zooms), uniform per channel."""
import numpy as np
import torch

from .. import ops


def _resize(vol, new_shape, order):
    """skimage.transform.resize(vol, new_shape, order, mode='edge', anti_aliasing=False) (clip=True) on a CUDA tensor."""
    out = ops.resize_volume(vol, new_shape, order)
    if order != 0:
        out = torch.clamp(out, vol.min().double(), vol.max().double())
    return out


def augment_discrete_linear_downsampling(data_sample, zoom_range=(1 / 6, 1 / 4, 1 / 2), zoom_axes_invidually=False, p=0.2,
                                         channels=None, order_downsample=1, order_upsample=0, ignore_axes=None,
                                         device="cuda"):
    """Reference: augment_discrete_linear_downsampling_scipy (discrete_downsampling.py:8-36).  data_sample [C,X,Y,Z]:
    numpy array (modified in place and returned, like the reference) or CUDA tensor (returned as a new tensor)."""
    if not isinstance(zoom_range, (list, tuple, np.ndarray)):
        zoom_range = [zoom_range]
    is_np = isinstance(data_sample, np.ndarray)
    shp = np.array(data_sample.shape[1:])
    zooms = np.random.choice(zoom_range, 3, replace=True) if zoom_axes_invidually else np.random.choice(zoom_range, 1)
    target_shape = np.round(shp * zooms).astype(int)
    if ignore_axes is not None:
        for i in ignore_axes:
            target_shape[i] = shp[i]
    if channels is None:
        channels = list(range(data_sample.shape[0]))
    out = data_sample if is_np else data_sample.clone()
    for c in channels:
        if np.random.uniform() < p:
            vol = (torch.from_numpy(np.ascontiguousarray(data_sample[c], dtype=np.float64)).to(device) if is_np
                   else data_sample[c].double())
            down = _resize(vol, [int(t) for t in target_shape], order_downsample)
            up = _resize(down, [int(s) for s in shp], order_upsample)
            if is_np:
                out[c] = up.cpu().numpy()
            else:
                out[c] = up.to(out.dtype)
    return out


class SimulateDiscreteLowResolutionTransform:
    """Same constructor arguments and call convention as the reference class (batchgenerators AbstractTransform style:
    `transform(**data_dict)` returns the dict)."""

    def __init__(self, zoom_range=(1 / 6, 1 / 4, 1 / 2), zoom_axes_invidually=False, per_channel=False, p_per_channel=1,
                 channels=None, order_downsample=1, order_upsample=0, data_key="data", p_per_sample=1, ignore_axes=None):
        self.order_upsample = order_upsample
        self.order_downsample = order_downsample
        self.channels = channels
        self.per_channel = per_channel
        self.p_per_channel = p_per_channel
        self.p_per_sample = p_per_sample
        self.data_key = data_key
        self.zoom_range = zoom_range
        self.zoom_axes_invidually = zoom_axes_invidually
        self.ignore_axes = ignore_axes

    def __call__(self, **data_dict):
        data = data_dict[self.data_key]
        for b in range(len(data)):
            if np.random.uniform() < self.p_per_sample:
                data[b] = augment_discrete_linear_downsampling(
                    data[b], zoom_range=self.zoom_range, zoom_axes_invidually=self.zoom_axes_invidually,
                    p=self.p_per_channel, channels=self.channels, order_downsample=self.order_downsample,
                    order_upsample=self.order_upsample, ignore_axes=self.ignore_axes)
        return data_dict
