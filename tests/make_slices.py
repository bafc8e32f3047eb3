"""Slicing of the gradients stored in tests/golden/full_*.npz (shared by the generator and the GPU test)."""
GRAD_SLICES = {
    "encoder.stages.0.0.convs.0.conv.weight": (slice(None), slice(None)),
    "encoder.stages.0.0.convs.1.conv.weight": (slice(None, None, 2), slice(None, None, 2)),
    "encoder.stages.1.0.convs.0.conv.weight": (slice(None, None, 4), slice(None, None, 4)),
    "encoder.stages.4.0.convs.1.conv.weight": (slice(None, None, 16), slice(None, None, 16)),
    "decoder.stages.0.convs.0.conv.weight": (slice(None, None, 16), slice(None, None, 16)),
    "decoder.stages.3.convs.0.conv.weight": (slice(None, None, 2), slice(None, None, 4)),
    "decoder.stages.3.convs.1.conv.weight": (slice(None, None, 2), slice(None, None, 2)),
    "decoder.transpconvs.0.weight": (slice(None, None, 16), slice(None, None, 16)),
    "decoder.transpconvs.3.weight": (slice(None), slice(None)),
    "decoder.seg_layers.3.weight": (slice(None), slice(None)),
    "decoder.seg_layers.3.bias": (slice(None),),
}
