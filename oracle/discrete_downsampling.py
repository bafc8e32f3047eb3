"""SimulateDiscreteLowResolutionTransform, CPU restatement (oracle; test infrastructure).

Follows /root/reference/dg_tta/pretraining/discrete_downsampling.py:8-72.  Its only arithmetic is
skimage.transform.resize(order, mode='edge', anti_aliasing=False) [third-party scikit-image, not installed here], restated
as scipy.ndimage.zoom(order, mode='nearest', grid_mode=True) + clip to the input range (oracle/preprocessing.resize).
The reference module itself cannot be imported here (it imports batchgenerators and skimage): parity unpinned; draw
order as written there (uniform per sample, choice of zooms, uniform per channel on numpy's global generator)."""
import numpy as np

from .preprocessing import resize


def augment(data_sample, zoom_range=(1 / 6, 1 / 4, 1 / 2), zoom_axes_invidually=False, p=0.2, channels=None,
            order_downsample=1, order_upsample=0, ignore_axes=None):
    if not isinstance(zoom_range, (list, tuple, np.ndarray)):
        zoom_range = [zoom_range]
    shp = np.array(data_sample.shape[1:])
    zooms = np.random.choice(zoom_range, 3, replace=True) if zoom_axes_invidually else np.random.choice(zoom_range, 1)
    target_shape = np.round(shp * zooms).astype(int)
    if ignore_axes is not None:
        for i in ignore_axes:
            target_shape[i] = shp[i]
    if channels is None:
        channels = list(range(data_sample.shape[0]))
    for c in channels:
        if np.random.uniform() < p:
            down = resize(data_sample[c].astype(float), target_shape, order_downsample)
            data_sample[c] = resize(down, shp, order_upsample)
    return data_sample


def transform(data, p_per_sample=1, **kw):
    for b in range(len(data)):
        if np.random.uniform() < p_per_sample:
            data[b] = augment(data[b], **kw)
    return data
