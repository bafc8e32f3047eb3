"""Minimal NIfTI-1 reader / writer (single-file .nii / .nii.gz; reads either byte order, writes little endian) for label maps and images.

The reference reads target cases and writes predictions through nnU-Net's SimpleITKIO [3P nnunetv2==2.2.1]
(dg_tta/tta/tta.py:411-446); neither SimpleITK nor nibabel is a dependency here, so this module restates the published
NIfTI-1 layout (348-byte header, vox_offset, datatype / bitpix, pixdim, qform / sform) for the interchange the TTA
workflow needs: read a segmentation or image volume with its geometry, write a segmentation with the same geometry.
Arrays are returned in (z, y, x) order like SimpleITK's GetArrayFromImage (NIfTI stores x fastest)."""
import gzip
import struct
from pathlib import Path

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


# (offset, struct code, count) of every multi-byte field of the 348-byte NIfTI-1 header; the rest are chars
_HDR_FIELDS = ((0, "i", 1), (32, "i", 1), (36, "h", 1), (40, "h", 8), (56, "f", 3), (68, "h", 4), (76, "f", 11), (120, "h", 1),
               (124, "f", 4), (140, "i", 2), (252, "h", 2), (256, "f", 18))


def _swap_header(raw):
    """The big-endian 348-byte header `raw` as little-endian bytes."""
    out = bytearray(raw)
    for off, code, n in _HDR_FIELDS:
        struct.pack_into(f"<{n}{code}", out, off, *struct.unpack_from(f">{n}{code}", raw, off))
    return bytes(out)


def read_nifti(path):
    """Returns (array [z,y,x] (or [t,z,y,x]), header dict with 'pixdim' (x,y,z spacing), 'affine' 4x4, 'raw' bytes)."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352 or 348 not in (struct.unpack("<i", raw[:4])[0], struct.unpack(">i", raw[:4])[0]):
        raise ValueError(f"{path}: not a NIfTI-1 file")
    bo = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"          # byte order of the file (round 5: big endian is read too)
    if raw[344:348] not in (b"n+1\0",):
        raise ValueError(f"{path}: only single-file NIfTI-1 (magic n+1) is supported")
    dim = struct.unpack(bo + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(bo + "hh", raw[70:74])
    pixdim = struct.unpack(bo + "8f", raw[76:108])
    vox_offset = int(struct.unpack(bo + "f", raw[108:112])[0])
    slope, inter = struct.unpack(bo + "ff", raw[112:120])
    if datatype not in _DTYPES:
        raise ValueError(f"{path}: unsupported NIfTI datatype {datatype}")
    nd = dim[0]
    shape = [int(d) for d in dim[1:1 + nd]]
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(bo)
    n = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=n, offset=vox_offset).reshape(shape[::-1])     # x fastest -> [.., z, y, x]
    if slope not in (0.0, 1.0) or inter != 0.0:
        data = data.astype(np.float32) * (slope if slope != 0.0 else 1.0) + inter
    qform_code, sform_code = struct.unpack(bo + "hh", raw[252:256])
    affine = np.eye(4, dtype=np.float64)
    if sform_code > 0:
        affine[:3] = np.array(struct.unpack(bo + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    else:
        affine[0, 0], affine[1, 1], affine[2, 2] = pixdim[1], pixdim[2], pixdim[3]
        if qform_code > 0:
            affine[:3, 3] = struct.unpack(bo + "3f", raw[268:280])
    hdr = {"pixdim": tuple(float(p) for p in pixdim[1:4]), "affine": affine, "shape_xyz": shape[:3]}
    # reused by write_nifti so that a prediction keeps the case's sform / qform, origin and orientation; a big-endian header
    # is byte-swapped field by field into the little-endian form the writer emits
    hdr["raw"] = bytes(raw[:348]) if bo == "<" else _swap_header(raw[:348])
    return np.ascontiguousarray(data.astype(data.dtype.newbyteorder("="))), hdr


def write_nifti(path, array, header=None, spacing=(1.0, 1.0, 1.0)):
    """Writes array [z,y,x] with the geometry of `header` (as returned by read_nifti) or an axis-aligned `spacing`."""
    arr = np.ascontiguousarray(array)
    if arr.dtype.name not in _CODES:
        arr = arr.astype(np.int16 if np.issubdtype(arr.dtype, np.integer) else np.float32)
    if arr.ndim != 3:
        raise ValueError("write_nifti expects a 3-D [z,y,x] array")
    if header is not None and "raw" not in header:
        spacing, header = tuple(header.get("pixdim", spacing)), None
    hdr = bytearray(header["raw"]) if header is not None else bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    z, y, x = arr.shape
    struct.pack_into("<8h", hdr, 40, 3, x, y, z, 1, 1, 1, 1)
    struct.pack_into("<hh", hdr, 70, _CODES[arr.dtype.name], arr.dtype.itemsize * 8)
    if header is None:
        struct.pack_into("<8f", hdr, 76, 1.0, float(spacing[0]), float(spacing[1]), float(spacing[2]), 1.0, 1.0, 1.0, 1.0)
        struct.pack_into("<hh", hdr, 252, 0, 1)        # sform only
        aff = np.diag([spacing[0], spacing[1], spacing[2]]).astype(np.float32)
        rows = np.concatenate([aff, np.zeros((3, 1), np.float32)], axis=1)
        struct.pack_into("<12f", hdr, 280, *rows.reshape(-1))
        hdr[123] = 2                                   # xyzt_units: mm
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<ff", hdr, 112, 1.0, 0.0)
    hdr[344:348] = b"n+1\0"
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")
        f.write(arr.astype(arr.dtype.newbyteorder("<")).tobytes())
