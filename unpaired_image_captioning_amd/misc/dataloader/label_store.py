"""The label file of the reference (`opt.input_label_h5`, written by scripts/prepro_labels.py:160-164 with h5py: `labels`
uint32 [M, L], `label_start_ix` / `label_end_ix` uint32 [n_images] 1-indexed, `label_length` uint32 [M]) and the NMT corpus
file (`opt.input_nmt_h5`, scripts/prepro_aic_nmt.py:435-448).

The reference opens them with h5py (P/misc/dataloader/dataloader.py:66,76).  h5py is not part of this image, so the same
files are read through whichever of these is there, in this order: h5py; the HDF5 C library itself through ctypes
(`libhdf5.so`, found on the loader path, in $UIC_HDF5_LIB, or in a conda prefix); and, for data sets converted
beforehand, an `.npz` holding arrays of the same names.  `write_hdf5` writes such a file with the C library (used by the
tests and by anyone building a data set without h5py).
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

LABEL_NAMES = ("labels", "label_start_ix", "label_end_ix", "label_length")
NMT_NAMES = tuple("%s_%s_label%s" % (s, side, suf) for s in ("train", "valid") for side in ("src", "tgt") for suf in ("", "_length"))


def open_label_store(path, names=LABEL_NAMES):
    """dict name -> numpy array (whole data sets in host memory, as the reference's driver='core')."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.endswith(".npz"):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, "r") as f:
            return {k: f[k][()] for k in names if k in f}
    return Hdf5Library().read(path, names)


class Hdf5Library(object):
    """The few calls of the HDF5 C API needed to read / write whole numeric data sets of the root group."""

    _NATIVE = {("i", 1, 1): ("H5T_NATIVE_INT8_g", np.int8), ("i", 1, 0): ("H5T_NATIVE_UINT8_g", np.uint8),
               ("i", 2, 1): ("H5T_NATIVE_INT16_g", np.int16), ("i", 2, 0): ("H5T_NATIVE_UINT16_g", np.uint16),
               ("i", 4, 1): ("H5T_NATIVE_INT32_g", np.int32), ("i", 4, 0): ("H5T_NATIVE_UINT32_g", np.uint32),
               ("i", 8, 1): ("H5T_NATIVE_INT64_g", np.int64), ("i", 8, 0): ("H5T_NATIVE_UINT64_g", np.uint64),
               ("f", 4, 0): ("H5T_NATIVE_FLOAT_g", np.float32), ("f", 8, 0): ("H5T_NATIVE_DOUBLE_g", np.float64)}

    def __init__(self):
        cands = [os.environ.get("UIC_HDF5_LIB"), ctypes.util.find_library("hdf5")]
        for prefix in (os.environ.get("CONDA_PREFIX"), "/opt/conda", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/lib/x86_64-linux-gnu"):
            if prefix:
                cands += sorted(glob.glob(os.path.join(prefix, "lib", "libhdf5.so*")) + glob.glob(os.path.join(prefix, "libhdf5*.so*")))
        self.lib = None
        for c in cands:
            if not c:
                continue
            try:
                self.lib = C.CDLL(c)
                break
            except OSError:
                continue
        if self.lib is None:
            raise ImportError("reading an HDF5 label file needs h5py or libhdf5.so (set UIC_HDF5_LIB), or convert the file to "
                              ".npz with the same array names")
        L = self.lib
        hid = C.c_int64
        for name, res, args in (("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]), ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]),
                                ("H5Fclose", C.c_int, [hid]), ("H5Lexists", C.c_int, [hid, C.c_char_p, hid]),
                                ("H5Dopen2", hid, [hid, C.c_char_p, hid]), ("H5Dclose", C.c_int, [hid]),
                                ("H5Dget_space", hid, [hid]), ("H5Dget_type", hid, [hid]),
                                ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
                                ("H5Sget_simple_extent_dims", C.c_int, [hid, C.c_void_p, C.c_void_p]),
                                ("H5Sclose", C.c_int, [hid]), ("H5Tget_class", C.c_int, [hid]), ("H5Tget_size", C.c_size_t, [hid]),
                                ("H5Tget_sign", C.c_int, [hid]), ("H5Tclose", C.c_int, [hid]),
                                ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
                                ("H5Screate_simple", hid, [C.c_int, C.c_void_p, C.c_void_p]),
                                ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]),
                                ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        if L.H5open() < 0:
            raise OSError("H5open failed")

    def _native(self, kind, size, signed):
        sym, dt = self._NATIVE[(kind, size, signed)]
        return C.c_int64.in_dll(self.lib, sym).value, dt

    def read(self, path, names):
        L = self.lib
        f = L.H5Fopen(path.encode(), 0, 0)                              # H5F_ACC_RDONLY, H5P_DEFAULT
        if f < 0:
            raise OSError("cannot open %s as HDF5" % path)
        out = {}
        try:
            for name in names:
                if L.H5Lexists(f, name.encode(), 0) <= 0:
                    continue
                d = L.H5Dopen2(f, name.encode(), 0)
                if d < 0:
                    raise OSError("cannot open data set %s of %s" % (name, path))
                sp, ty = L.H5Dget_space(d), L.H5Dget_type(d)
                nd = L.H5Sget_simple_extent_ndims(sp)
                dims = (C.c_uint64 * max(nd, 1))()
                L.H5Sget_simple_extent_dims(sp, dims, None)
                cls = L.H5Tget_class(ty)                                # 0 = H5T_INTEGER, 1 = H5T_FLOAT
                if cls not in (0, 1):
                    raise TypeError("data set %s of %s is not numeric" % (name, path))
                size = int(L.H5Tget_size(ty))
                signed = int(L.H5Tget_sign(ty) == 1) if cls == 0 else 0  # H5T_SGN_2
                mem, dt = self._native("i" if cls == 0 else "f", size, signed)
                arr = np.empty(tuple(int(dims[i]) for i in range(nd)), dtype=dt)
                if L.H5Dread(d, mem, 0, 0, 0, arr.ctypes.data) < 0:     # H5S_ALL, H5S_ALL, H5P_DEFAULT
                    raise OSError("cannot read data set %s of %s" % (name, path))
                L.H5Tclose(ty); L.H5Sclose(sp); L.H5Dclose(d)
                out[name] = arr
        finally:
            L.H5Fclose(f)
        return out

    def write(self, path, arrays):
        L = self.lib
        f = L.H5Fcreate(path.encode(), 2, 0, 0)                         # H5F_ACC_TRUNC
        if f < 0:
            raise OSError("cannot create %s" % path)
        try:
            for name, a in arrays.items():
                a = np.ascontiguousarray(a)
                kind = "f" if a.dtype.kind == "f" else "i"
                mem, _ = self._native(kind, a.dtype.itemsize, int(a.dtype.kind == "i") if kind == "i" else 0)
                dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
                sp = L.H5Screate_simple(a.ndim, dims, None)
                d = L.H5Dcreate2(f, name.encode(), mem, sp, 0, 0, 0)
                if d < 0 or L.H5Dwrite(d, mem, 0, 0, 0, a.ctypes.data) < 0:
                    raise OSError("cannot write data set %s of %s" % (name, path))
                L.H5Dclose(d); L.H5Sclose(sp)
        finally:
            L.H5Fclose(f)


def write_hdf5(path, arrays):
    """dict name -> numeric numpy array, each written as one contiguous data set of the root group (what h5py's
    create_dataset(name, data=...) of the reference's prepro scripts produces)."""
    Hdf5Library().write(path, arrays)
