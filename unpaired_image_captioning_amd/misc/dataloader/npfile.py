"""Per-image feature files (.npy, and .npz as np.savez writes them: a zip of STORED .npy members) read with one read()
and one header parse instead of np.load's zipfile machinery (≈ 0.3 ms of interpreter time per file, the loader's
bottleneck at 3 files per image).  Anything unexpected -- compressed members, another member first, pickled objects --
goes through np.load."""
import ast
import struct

import numpy as np


def _npy_view(buf, off):
    """ndarray view of the .npy image that starts at buf[off] (numpy/lib/format.py: magic, version, header dict)."""
    if buf[off:off + 6] != b"\x93NUMPY":
        return None
    major = buf[off + 6]
    if major == 1:
        hlen, = struct.unpack_from("<H", buf, off + 8)
        start = off + 10
    elif major in (2, 3):
        hlen, = struct.unpack_from("<I", buf, off + 8)
        start = off + 12
    else:
        return None
    header = ast.literal_eval(buf[start:start + hlen].decode("latin1" if major < 3 else "utf8"))
    dtype = np.dtype(header["descr"])
    if dtype.hasobject or header["fortran_order"]:
        return None
    shape = tuple(header["shape"])
    count = int(np.prod(shape)) if shape else 1
    return np.frombuffer(buf, dtype=dtype, count=count, offset=start + hlen).reshape(shape)


def load_array(path, member="feat"):
    """The array of a .npy file, or member `member` of an .npz file (read-only view of the file's bytes)."""
    with open(path, "rb") as f:
        buf = f.read()
    arr = None
    if buf[:4] == b"PK\x03\x04":                                        # zip local file header
        method, = struct.unpack_from("<H", buf, 8)
        nlen, xlen = struct.unpack_from("<HH", buf, 26)
        if method == 0 and buf[30:30 + nlen] == (member + ".npy").encode():
            arr = _npy_view(buf, 30 + nlen + xlen)
        if arr is None:
            with np.load(path) as z:
                return z[member]
    else:
        arr = _npy_view(buf, 0)
        if arr is None:
            return np.load(path)
    return arr
