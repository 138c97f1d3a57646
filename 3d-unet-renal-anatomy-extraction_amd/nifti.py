"""Minimal NIfTI-1 single-file reader / writer (.nii, .nii.gz) in numpy.

The reference reads and writes its cases with nibabel (data.py:36-47 `nib.load(...).get_fdata()`, :82-97
`nib.save(nib.Nifti1Pair(array, affine), path)`).  nibabel is not installed in this image; when it is importable this
module hands the call to it, otherwise it implements the part of the format those call sites use: the 348-byte
NIfTI-1 header (little or big endian), the scalar data types below, scl_slope / scl_inter scaling (what `get_fdata`
applies), and the voxel-to-world affine from the sform rows (sform_code > 0), else the qform quaternion, else pixdim.
Arrays are Fortran-ordered on disk, exactly as the standard prescribes.
"""
import gzip
import struct

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import nibabel as _nib
except Exception:
    _nib = None

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quaternion_affine(b, c, d, qfac, pixdim, offset):
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    rot = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                    [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                    [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    zooms = np.array([pixdim[1], pixdim[2], pixdim[3] * (-1.0 if qfac < 0 else 1.0)])
    aff = np.eye(4)
    aff[:3, :3] = rot * zooms
    aff[:3, 3] = offset
    return aff


def load(path):
    """-> (array float64 scaled like nibabel's get_fdata(), affine 4x4 float64, header dict)."""
    if _nib is not None:  # pragma: no cover
        img = _nib.load(str(path))
        return img.get_fdata(), np.asarray(img.affine, dtype=np.float64), {"nibabel": True}
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise ValueError("%s: too short for a NIfTI-1 file" % path)
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348:
        raise ValueError("%s: not a NIfTI-1 header (sizeof_hdr != 348)" % path)
    if raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError("%s: bad NIfTI magic %r" % (path, raw[344:348]))
    if raw[344:347] == b"ni1":
        raise ValueError("%s: header/image pairs (.hdr/.img) are not supported, use a single .nii(.gz) file" % path)
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "hh", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "fff", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "hh", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
    srow = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError("%s: unsupported NIfTI datatype code %d" % (path, datatype))
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    count = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dt, count=count, offset=int(vox_offset)).reshape(shape, order="F")
    arr = data.astype(np.float64)
    if slope not in (0.0,) and not np.isnan(slope) and (slope != 1.0 or inter != 0.0):
        arr = arr * float(slope) + float(inter)
    if sform_code > 0:
        affine = np.vstack([srow, [0, 0, 0, 1]])
    elif qform_code > 0:
        affine = _quaternion_affine(qb, qc, qd, pixdim[0], pixdim, (qx, qy, qz))
    else:
        affine = np.diag([pixdim[1] or 1.0, pixdim[2] or 1.0, pixdim[3] or 1.0, 1.0])
    return np.asarray(arr, order="C"), affine, {"datatype": datatype, "pixdim": pixdim, "dim": dim}


def save(array, affine, path):
    """Write `array` (its own dtype) with the voxel-to-world `affine` as sform + qform-less NIfTI-1 (.nii / .nii.gz)."""
    array = np.asarray(array)
    if _nib is not None:  # pragma: no cover
        _nib.save(_nib.Nifti1Image(array, np.asarray(affine)), str(path))
        return
    dt = np.dtype(array.dtype)
    if dt == np.bool_:
        array, dt = array.astype(np.uint8), np.dtype(np.uint8)
    if dt.newbyteorder("=") not in _CODES and dt not in _CODES:
        raise ValueError("nifti.save: unsupported dtype %s" % dt)
    code = _CODES[np.dtype(dt.type)]
    if array.ndim > 7:
        raise ValueError("nifti.save: at most 7 dimensions")
    affine = np.asarray(affine, dtype=np.float64)
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    dims = [array.ndim] + list(array.shape) + [1] * (7 - array.ndim)
    struct.pack_into("<8h", hdr, 40, *dims)
    struct.pack_into("<hh", hdr, 70, code, dt.itemsize * 8)
    zooms = [float(np.linalg.norm(affine[:3, i])) for i in range(3)]
    struct.pack_into("<8f", hdr, 76, 1.0, *(zooms + [1.0] * 4))
    struct.pack_into("<fff", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2                                               # xyzt_units: millimetres
    struct.pack_into("<hh", hdr, 252, 0, 2)                    # qform unknown, sform aligned
    struct.pack_into("<12f", hdr, 280, *affine[:3].reshape(-1))
    hdr[344:348] = b"n+1\x00"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(np.asarray(array, dtype=dt.newbyteorder("<")).tobytes(order="F"))
