"""ctypes binding of libdgtta_hip.so (the C ABI declared in include/dgtta.h).

There is NO CPU fallback: if the library cannot be loaded every op raises.  The library is built in-tree by
`python dg_tta_amd/build.py` (hipcc, gfx950) so that it travels with the repository snapshot.
"""
import ctypes as C
import os
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libdgtta_hip.so"

_c_float_p = C.POINTER(C.c_float)
P, I, I64, F, SZ = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/dgtta.h one to one
SIGNATURES = {
    "dgtta_version": (I, []),
    "dgtta_last_error": (C.c_char_p, []),
    "dgtta_reload_env": (I, []),
    "dgtta_mind3d_ws_bytes": (SZ, [I, I, I, I]),
    "dgtta_mind3d_fwd": (I, [P, P, F, I, C.POINTER(F), I, P, I, I, I, P, SZ, I, I, I, I, P]),
    "dgtta_gin_ws_bytes": (SZ, [I, I, I, I]),
    "dgtta_gin_chain_fwd": (I, [P, P, C.POINTER(I), C.POINTER(P), C.POINTER(P), P, P, SZ, I, I, I, I, P]),
    "dgtta_affine_warp3d_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P, P]),
    "dgtta_affine_warp3d_bwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_softdice_ws_bytes": (SZ, [I, I, I64]),
    "dgtta_softdice_fwd": (I, [P, P, P, P, P, SZ, I, I, I64, I, I, I, P]),
    "dgtta_softdice_bwd": (I, [P, P, P, P, P, F, P, I, I, I64, I, I, P]),
    "dgtta_softdice_bwd_t": (I, [P, P, P, P, P, F, P, I, I, I64, I, I, I, P]),
    "dgtta_softdice_probs_fwd": (I, [P, P, P, P, SZ, I, I, I64, I64, I64, I64, P]),
    "dgtta_softdice_probs_bwd": (I, [P, P, P, P, P, P, I, I, I64, I64, I64, I64, P]),
    "dgtta_dice_ce_ws_bytes": (SZ, [I, I, I64]),
    "dgtta_dice_ce_fwd": (I, [P, I, P, P, P, P, SZ, I, I, I64, F, I, P]),
    "dgtta_dice_ce_bwd": (I, [P, I, P, P, F, P, P, I, I, I, I64, P]),
    "dgtta_adamw_step": (I, [C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(I64), I, F, F, F, F, F,
                             I, F, P, P]),
    "dgtta_grads_nonfinite": (I, [C.POINTER(P), C.POINTER(I64), I, P, P]),
    "dgtta_conv3d_packed_bytes": (SZ, [I, I, I]),
    "dgtta_conv3d_pack_weights": (I, [P, P, I, I, I, I, I, P]),
    "dgtta_conv3d_stats_bytes": (SZ, [I, I, I, I, I]),
    "dgtta_conv3d_k3_fwd": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_conv3d_k3_dgrad": (I, [P, I, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_conv3d_wgrad_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "dgtta_conv3d_wgrad_split_ws_bytes": (SZ, [I, I, I, I, I, I, I]),
    "dgtta_conv3d_k3_wgrad": (I, [P, I, P, I, P, P, P, SZ, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_conv3d_k3_blocked_supported": (I, [I, I, I, I, I, I, I]),
    "dgtta_conv3d_k3_fwd_blocked": (I, [P, I64, P, P, P, I, P, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_conv3d_k3_wgrad_blocked": (I, [P, I64, P, I, P, P, P, SZ, I, I, I, I, I, I, I, I, P]),
    "dgtta_instnorm_ws_bytes": (SZ, [I, I, I64]),
    "dgtta_instnorm_lrelu_fwd": (I, [P, I, P, P, P, P, P, I, P, SZ, I, I, I64, F, F, I, P]),
    "dgtta_instnorm_lrelu_bwd": (I, [P, I, P, I, P, P, P, P, I, P, P, P, SZ, I, I, I64, F, I, I, P]),
    "dgtta_conv3d_k3_dgrad_gstats": (I, [P, I, P, P, I, I, I, I, I, I, I, I, I, P, I, P, P, P, F, P, SZ, C.POINTER(I), I, I, P]),
    "dgtta_instnorm_lrelu_bwd_gstats": (I, [P, I, P, I, P, P, P, P, I, P, P, P, P, SZ, I, I, I64, F, I, I, P]),
    "dgtta_convT3d_fwd_ws_bytes": (SZ, [I, I, I]),
    "dgtta_convT3d_k2s2_fwd": (I, [P, I, P, P, P, I, P, SZ, I, I, I, I, I, I, I, I, P]),
    "dgtta_convT3d_bwd_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "dgtta_convT3d_bwd_split_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "dgtta_convT3d_k2s2_bwd": (I, [P, I, P, I, P, P, I, P, P, P, SZ, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_seghead_fwd": (I, [P, I, P, P, P, I, P, I, I, I, I, I64, I, P]),
    "dgtta_seghead_bwd_ws_bytes": (SZ, [I, I, I, I64]),
    "dgtta_seghead_bwd": (I, [P, I, P, I, P, P, I, P, I, P, P, P, SZ, I, I, I64, I, I, P]),
    "dgtta_seghead_warp_supported": (I, [P, I, I, I, I, I, I, I]),
    "dgtta_seghead_warp_bwd_ws_bytes": (SZ, [I, I, I, I, I, I]),
    "dgtta_seghead_warp_fwd": (I, [P, P, P, P, I, P, P, I, I, I, I, I, I, I, P]),
    "dgtta_seghead_warp_bwd": (I, [P, P, P, P, P, P, I, P, P, P, P, SZ, I, I, I, I, I, I, I, I, P]),
    "dgtta_seghead_warp_bwd_g16": (I, [P, P, P, P, P, P, I, P, P, P, P, SZ, I, I, I, I, I, I, I, I, P]),
    "dgtta_ncdhw_to_ndhwc": (I, [P, P, I, I, I64, I, I, P]),
    "dgtta_ndhwc_to_ncdhw": (I, [P, P, I, I, I64, I, I, P]),
    "dgtta_argmax_dice": (I, [P, I, I, P, P, P, I, I64, P]),
    "dgtta_resample_axis_ws_bytes": (SZ, [I64, I, I64, I]),
    "dgtta_resample_axis": (I, [P, P, P, SZ, I64, I, I, I64, I, P]),
    "dgtta_window_accumulate": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_seghead_window_accumulate": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_logits_chunk_f64": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_window_accumulate_t": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_seghead_window_accumulate_t": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_argmax_rows": (I, [P, I, I, I64, P, P]),
    "dgtta_feature_window_accumulate": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_feature_window_accumulate_norm": (I, [P, P, P, P, F, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_feature_window_accumulate_multi": (I, [P, P, P, I, P, P, F, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_feature_head_argmax": (I, [P, I64, P, P, P, I, I, I, I64, P, P]),
    "dgtta_feature_logits_chunk_f64": (I, [P, I64, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_logits_chunk_f64_t": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "dgtta_argmax_merge_f64": (I, [P, I64, I, I, P, P, I, P]),
}

_lib = None
_load_error = None


class DgttaError(RuntimeError):
    pass


def load():
    """Loads the shared library once; raises DgttaError (never falls back) when it is missing."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if _load_error is not None:
        raise DgttaError(_load_error)
    path = os.environ.get("DGTTA_LIB", str(LIB_PATH))
    # PyTorch-ROCm ships its own libamdhip64.so.7; import it first so that this library binds to the SAME HIP runtime
    # instance (same streams / allocations) instead of pulling in /opt/rocm's copy as a second runtime.
    import torch  # noqa: F401
    try:
        lib = C.CDLL(path)
    except OSError as e:
        _load_error = (f"libdgtta_hip.so could not be loaded from {path} ({e}). Build it with "
                       f"`python dg_tta_amd/build.py`; dg_tta_amd has no CPU fallback.")
        raise DgttaError(_load_error) from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dgtta_last_error().decode(errors="replace")
        raise DgttaError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_of(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DgttaError("dg_tta_amd ops run on the MI355X only: got a CPU tensor (there is no CPU fallback)")
