"""Image file I/O of the TTA workflow beyond NIfTI (round 5).

The reference reads target cases and writes predictions through nnU-Net's SimpleITKIO [3P nnunetv2==2.2.1]
(dg_tta/tta/tta.py:404-446, dg_tta/tta/nnunet_utils.py:170-204): whatever SimpleITK reads.  SimpleITK is not a dependency here;
this module restates the published layouts of the volume formats nnU-Net datasets are shipped in besides NIfTI-1 (`nifti_io.py`):

* NRRD (`.nrrd`, attached data; `NRRD000x` text header: type, dimension, sizes, space directions | spacings, space origin,
  encoding raw | gzip, endian),
* MetaImage (`.mha`, and `.mhd` with its `ElementDataFile`; NDims, DimSize, ElementSpacing, Offset, TransformMatrix, ElementType,
  BinaryDataByteOrderMSB, CompressedData = zlib).

`read_image` returns `(array [z, y, x], header)` like `nifti_io.read_nifti` (SimpleITK's GetArrayFromImage order; the files store x
fastest), with `header["pixdim"]` = (x, y, z) spacing, `header["format"]`, `header["ext"]` (the ending a prediction of this case is
written with, as SimpleITK's writer picks the format from the file name) and what the format's writer needs to reproduce the
geometry.  DICOM series and the other SimpleITK formats stay out of scope (the reader says so)."""
import gzip
import re
import zlib
from pathlib import Path

import numpy as np

from .nifti_io import read_nifti, write_nifti

EXTENSIONS = (".nii.gz", ".nii", ".nrrd", ".mha", ".mhd")


def extension_of(path):
    n = Path(path).name.lower()
    for e in EXTENSIONS:
        if n.endswith(e):
            return e
    return None


def is_image_file(path):
    return extension_of(path) is not None


# ------------------------------------------------------------------------------------------------ NRRD
_NRRD_TYPES = {"signed char": "i1", "int8": "i1", "int8_t": "i1", "uchar": "u1", "unsigned char": "u1", "uint8": "u1", "uint8_t": "u1",
               "short": "i2", "short int": "i2", "signed short": "i2", "signed short int": "i2", "int16": "i2", "int16_t": "i2",
               "ushort": "u2", "unsigned short": "u2", "unsigned short int": "u2", "uint16": "u2", "uint16_t": "u2",
               "int": "i4", "signed int": "i4", "int32": "i4", "int32_t": "i4", "uint": "u4", "unsigned int": "u4", "uint32": "u4",
               "uint32_t": "u4", "longlong": "i8", "long long": "i8", "long long int": "i8", "signed long long": "i8",
               "signed long long int": "i8", "int64": "i8", "int64_t": "i8", "ulonglong": "u8", "unsigned long long": "u8",
               "unsigned long long int": "u8", "uint64": "u8", "uint64_t": "u8", "float": "f4", "double": "f8"}
_NRRD_NAMES = {"i1": "int8", "u1": "uint8", "i2": "int16", "u2": "uint16", "i4": "int32", "u4": "uint32", "i8": "int64", "u8": "uint64",
               "f4": "float", "f8": "double"}


def _vec(text):
    return [float(v) for v in text.strip().strip("()").split(",")]


def read_nrrd(path):
    raw = Path(path).read_bytes()
    if not raw.startswith(b"NRRD"):
        raise ValueError(f"{path}: not an NRRD file")
    end = raw.find(b"\n\n")
    sep = 2
    if end < 0 or (0 <= raw.find(b"\r\n\r\n") < end):
        end, sep = raw.find(b"\r\n\r\n"), 4
    if end < 0:
        raise ValueError(f"{path}: NRRD header without the terminating blank line")
    fields = {}
    for line in raw[:end].decode("ascii", "replace").splitlines()[1:]:
        if not line or line.startswith("#"):
            continue
        if ":=" in line:        # key / value pair, not a field
            continue
        k, _, v = line.partition(":")
        fields[k.strip().lower()] = v.strip()
    if "data file" in fields or "datafile" in fields:
        raise NotImplementedError(f"{path}: detached NRRD data files are not supported")
    if int(fields.get("dimension", "0")) != 3:
        raise ValueError(f"{path}: only 3-D NRRD volumes (dimension {fields.get('dimension')})")
    code = _NRRD_TYPES.get(fields["type"].lower())
    if code is None:
        raise ValueError(f"{path}: unsupported NRRD type {fields['type']}")
    sizes = [int(v) for v in fields["sizes"].split()]
    for skip in ("byte skip", "byteskip", "line skip", "lineskip"):
        if int(fields.get(skip, "0")) != 0:
            raise NotImplementedError(f"{path}: NRRD '{skip}: {fields[skip]}' is not supported")
    enc = fields.get("encoding", "raw").lower()
    body = raw[end + sep:]
    if enc in ("gzip", "gz"):
        body = gzip.decompress(body)
    elif enc != "raw":
        raise NotImplementedError(f"{path}: NRRD encoding {enc}")
    bo = ">" if fields.get("endian", "little").lower() == "big" else "<"
    data = np.frombuffer(body, dtype=np.dtype(bo + code), count=int(np.prod(sizes))).reshape(sizes[::-1])
    direction, origin = np.eye(3), np.zeros(3)
    if "space directions" in fields:
        vecs = re.findall(r"\([^)]*\)", fields["space directions"])
        direction = np.array([_vec(v) for v in vecs], dtype=np.float64).T          # columns = axis vectors
        spacing = tuple(float(np.linalg.norm(direction[:, i])) for i in range(3))
    elif "spacings" in fields:
        spacing = tuple(float(v) for v in fields["spacings"].split())
        direction = np.diag(spacing)
    else:
        spacing = (1.0, 1.0, 1.0)
    if "space origin" in fields:
        origin = np.array(_vec(fields["space origin"]), dtype=np.float64)
    hdr = {"pixdim": spacing, "format": "nrrd", "ext": ".nrrd", "direction": direction, "origin": origin,
           "space": fields.get("space", "left-posterior-superior"), "shape_xyz": sizes}
    return np.ascontiguousarray(data.astype(data.dtype.newbyteorder("="))), hdr


def write_nrrd(path, array, header=None, spacing=(1.0, 1.0, 1.0), compress=True):
    arr = np.ascontiguousarray(array)
    if arr.ndim != 3:
        raise ValueError("write_nrrd expects a 3-D [z,y,x] array")
    code = arr.dtype.kind + str(arr.dtype.itemsize)
    if code not in _NRRD_NAMES:
        arr = arr.astype(np.int16 if np.issubdtype(arr.dtype, np.integer) else np.float32)
        code = arr.dtype.kind + str(arr.dtype.itemsize)
    direction = np.asarray(header["direction"], dtype=np.float64) if header is not None and "direction" in header else np.diag(spacing)
    origin = np.asarray(header["origin"], dtype=np.float64) if header is not None and "origin" in header else np.zeros(3)
    z, y, x = arr.shape
    fmt = lambda v: "(" + ",".join(repr(float(a)) for a in v) + ")"        # noqa: E731
    lines = ["NRRD0004", f"type: {_NRRD_NAMES[code]}", "dimension: 3",
             f"space: {header.get('space', 'left-posterior-superior') if header else 'left-posterior-superior'}", f"sizes: {x} {y} {z}",
             "space directions: " + " ".join(fmt(direction[:, i]) for i in range(3)), "kinds: domain domain domain", "endian: little",
             f"encoding: {'gzip' if compress else 'raw'}", "space origin: " + fmt(origin)]
    body = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n\n").encode("ascii"))
        f.write(gzip.compress(body, 1) if compress else body)


# ------------------------------------------------------------------------------------------------ MetaImage
_MET = {"MET_CHAR": "i1", "MET_UCHAR": "u1", "MET_SHORT": "i2", "MET_USHORT": "u2", "MET_INT": "i4", "MET_UINT": "u4",
        "MET_LONG": "i4", "MET_ULONG": "u4", "MET_LONG_LONG": "i8", "MET_ULONG_LONG": "u8", "MET_FLOAT": "f4", "MET_DOUBLE": "f8"}
_MET_NAMES = {"i1": "MET_CHAR", "u1": "MET_UCHAR", "i2": "MET_SHORT", "u2": "MET_USHORT", "i4": "MET_INT", "u4": "MET_UINT",
              "i8": "MET_LONG_LONG", "u8": "MET_ULONG_LONG", "f4": "MET_FLOAT", "f8": "MET_DOUBLE"}


def read_metaimage(path):
    raw = Path(path).read_bytes()
    fields, pos = {}, 0
    while True:                                   # `key = value` lines up to and including ElementDataFile
        nl = raw.find(b"\n", pos)
        if nl < 0:
            raise ValueError(f"{path}: MetaImage header without ElementDataFile")
        line = raw[pos:nl].decode("ascii", "replace").strip()
        pos = nl + 1
        if "=" not in line:
            continue
        k, _, v = line.partition("=")
        fields[k.strip()] = v.strip()
        if k.strip() == "ElementDataFile":
            break
    if int(fields.get("NDims", "0")) != 3:
        raise ValueError(f"{path}: only 3-D MetaImage volumes (NDims {fields.get('NDims')})")
    if int(fields.get("ElementNumberOfChannels", "1")) != 1:
        raise NotImplementedError(f"{path}: multi-channel MetaImage elements")
    code = _MET.get(fields["ElementType"])
    if code is None:
        raise ValueError(f"{path}: unsupported ElementType {fields['ElementType']}")
    sizes = [int(v) for v in fields["DimSize"].split()]
    msb = fields.get("BinaryDataByteOrderMSB", fields.get("ElementByteOrderMSB", "False")).lower() == "true"
    data_file = fields["ElementDataFile"]
    if data_file == "LOCAL":
        body = raw[pos:]
    elif data_file.upper().startswith("LIST") or "%" in data_file:
        raise NotImplementedError(f"{path}: MetaImage slice lists are not supported")
    else:
        body = (Path(path).parent / data_file).read_bytes()
    if fields.get("CompressedData", "False").lower() == "true":
        body = zlib.decompress(body)
    hs = int(fields.get("HeaderSize", "0"))
    if hs < 0:          # -1: "the data are the LAST bytes of the file" - needs the element count against the file size
        raise NotImplementedError(f"{path}: MetaImage HeaderSize = {hs} is not supported")
    if hs > 0:
        body = body[hs:]
    data = np.frombuffer(body, dtype=np.dtype((">" if msb else "<") + code), count=int(np.prod(sizes))).reshape(sizes[::-1])
    spacing = tuple(float(v) for v in fields.get("ElementSpacing", fields.get("ElementSize", "1 1 1")).split())
    origin = np.array([float(v) for v in fields.get("Offset", fields.get("Position", fields.get("Origin", "0 0 0"))).split()])
    tm = fields.get("TransformMatrix", fields.get("Rotation", fields.get("Orientation")))
    rot = np.array([float(v) for v in tm.split()]).reshape(3, 3).T if tm else np.eye(3)      # rows of the file = axis directions
    hdr = {"pixdim": spacing, "format": "metaimage", "ext": ".mha", "direction": rot * np.array(spacing)[None, :], "origin": origin,
           "rotation": rot, "shape_xyz": sizes, "anatomical_orientation": fields.get("AnatomicalOrientation")}
    return np.ascontiguousarray(data.astype(data.dtype.newbyteorder("="))), hdr


def write_metaimage(path, array, header=None, spacing=(1.0, 1.0, 1.0), compress=True):
    arr = np.ascontiguousarray(array)
    if arr.ndim != 3:
        raise ValueError("write_metaimage expects a 3-D [z,y,x] array")
    code = arr.dtype.kind + str(arr.dtype.itemsize)
    if code not in _MET_NAMES:
        arr = arr.astype(np.int16 if np.issubdtype(arr.dtype, np.integer) else np.float32)
        code = arr.dtype.kind + str(arr.dtype.itemsize)
    if header is not None:
        spacing = tuple(header["pixdim"])
    rot = np.asarray(header["rotation"], dtype=np.float64) if header is not None and "rotation" in header else np.eye(3)
    origin = np.asarray(header["origin"], dtype=np.float64) if header is not None and "origin" in header else np.zeros(3)
    z, y, x = arr.shape
    body = arr.astype(arr.dtype.newbyteorder("<")).tobytes()
    if compress:
        body = zlib.compress(body, 1)
    num = lambda v: " ".join(repr(float(a)) for a in v)        # noqa: E731
    detached = str(path).lower().endswith(".mhd")
    data_name = Path(path).with_suffix(".zraw" if compress else ".raw").name if detached else "LOCAL"
    lines = ["ObjectType = Image", "NDims = 3", "BinaryData = True", "BinaryDataByteOrderMSB = False",
             f"CompressedData = {'True' if compress else 'False'}"]
    if compress:
        lines.append(f"CompressedDataSize = {len(body)}")
    lines += [f"TransformMatrix = {num(rot.T.reshape(-1))}", f"Offset = {num(origin)}", "CenterOfRotation = 0 0 0",
              f"ElementSpacing = {num(spacing)}", f"DimSize = {x} {y} {z}",
              f"AnatomicalOrientation = {(header or {}).get('anatomical_orientation') or 'RAI'}", f"ElementType = {_MET_NAMES[code]}",
              f"ElementDataFile = {data_name}"]
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        if not detached:
            f.write(body)
    if detached:
        (Path(path).parent / data_name).write_bytes(body)


# ------------------------------------------------------------------------------------------------ dispatch
def read_image(path):
    """(array [z,y,x], header) of a NIfTI / NRRD / MetaImage volume; header['pixdim'] = (x, y, z) spacing."""
    ext = extension_of(path)
    if ext in (".nii", ".nii.gz"):
        arr, hdr = read_nifti(path)
        hdr = dict(hdr, format="nifti", ext=ext)        # the case's real ending: labels and predictions are looked up by it
        return arr, hdr
    if ext == ".nrrd":
        return read_nrrd(path)
    if ext in (".mha", ".mhd"):
        arr, hdr = read_metaimage(path)
        hdr["ext"] = ext
        return arr, hdr
    raise NotImplementedError(f"{Path(path).name}: supported volume formats are {', '.join(EXTENSIONS)} (DICOM series and the other "
                              f"SimpleITK formats need SimpleITK, which is not a dependency)")


def write_image(path, array, header=None, spacing=(1.0, 1.0, 1.0)):
    """Writes [z,y,x] in the format of the file name, with the geometry of `header` when it comes from the same format (a
    header of another format carries over its spacing)."""
    ext = extension_of(path)
    same = header is not None and header.get("format") == {".nii": "nifti", ".nii.gz": "nifti", ".nrrd": "nrrd", ".mha": "metaimage",
                                                           ".mhd": "metaimage"}.get(ext)
    if header is not None and not same:
        spacing, header = tuple(header.get("pixdim", spacing)), None
    if ext in (".nii", ".nii.gz"):
        return write_nifti(path, array, header=header, spacing=spacing)
    if ext == ".nrrd":
        return write_nrrd(path, array, header=header, spacing=spacing)
    if ext in (".mha", ".mhd"):
        return write_metaimage(path, array, header=header, spacing=spacing)
    raise NotImplementedError(f"{Path(path).name}: cannot write this format")
