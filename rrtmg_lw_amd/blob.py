"""RRLWBLOB: the tiny named-array container every table in this package travels in.

Layout (little endian):

    char[8]  magic  "RRLWBLOB"
    u32      version (1)
    u32      nentries
    nentries x entry header (96 bytes each):
        char[48] name (NUL padded)
        u32      dtype   0 = float64, 1 = int32
        u32      ndim    (<= 6)
        u32[6]   dims    Fortran order: dims[0] is the fastest-varying axis
        u64      offset  byte offset of the payload from the start of the file
        u64      nbytes
    payloads, each 64-byte aligned, stored in Fortran (column-major) element order

The same file is read by the C oracle (oracle/rrlw_blob.h), the C++ product loader
(rrtmg_lw_amd/csrc/tables.hpp), the Fortran reference harness (oracle/ref_harness.f90) and this module.
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"RRLWBLOB"
VERSION = 1
_HDR = struct.Struct("<8sII")
_ENT = struct.Struct("<48sII6IQQ")
_DTYPES = {0: np.dtype("<f8"), 1: np.dtype("<i4")}
_CODES = {np.dtype("<f8"): 0, np.dtype("<i4"): 1}


def write_blob(path, arrays):
    """arrays: mapping name -> ndarray whose *numpy shape is the Fortran shape* (axis 0 fastest in the file)."""
    items = []
    for name, a in arrays.items():
        a = np.asarray(a)
        if a.dtype.kind == "f":
            a = a.astype("<f8")
        elif a.dtype.kind in "iu":
            a = a.astype("<i4")
        else:
            raise TypeError(f"{name}: unsupported dtype {a.dtype}")
        if a.ndim > 6:
            raise ValueError(f"{name}: ndim {a.ndim} > 6")
        if len(name.encode()) > 47:
            raise ValueError(f"name too long: {name}")
        items.append((name, a))
    off = _HDR.size + _ENT.size * len(items)
    heads, payloads = [], []
    for name, a in items:
        off = (off + 63) // 64 * 64
        data = np.asfortranarray(a).tobytes(order="F")
        dims = list(a.shape) + [1] * (6 - a.ndim)
        heads.append(_ENT.pack(name.encode(), _CODES[a.dtype], a.ndim, *dims, off, len(data)))
        payloads.append((off, data))
        off += len(data)
    with open(path, "wb") as f:
        f.write(_HDR.pack(MAGIC, VERSION, len(items)))
        for h in heads:
            f.write(h)
        for o, d in payloads:
            f.seek(o)
            f.write(d)


def read_blob(path):
    """Returns OrderedDict name -> ndarray with the Fortran shape (Fortran-contiguous)."""
    with open(path, "rb") as f:
        buf = f.read()
    magic, ver, n = _HDR.unpack_from(buf, 0)
    if magic != MAGIC or ver != VERSION:
        raise ValueError(f"{path}: not an RRLWBLOB v{VERSION} file")
    out = OrderedDict()
    for i in range(n):
        rec = _ENT.unpack_from(buf, _HDR.size + i * _ENT.size)
        name = rec[0].split(b"\0", 1)[0].decode()
        dt = _DTYPES[rec[1]]
        ndim = rec[2]
        dims = rec[3:3 + ndim]
        off, nbytes = rec[9], rec[10]
        a = np.frombuffer(buf, dtype=dt, count=nbytes // dt.itemsize, offset=off)
        out[name] = a.reshape(dims, order="F") if ndim else a.reshape(())
    return out
