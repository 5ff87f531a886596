"""A minimal ctypes binding to the REAL HDF5 C library (libhdf5), for the interop tests of the snapshot files
(tests/test_libhdf5_interop.py): open / create files, walk groups, read and write f64 datasets.  Test infrastructure only.

The library is looked for where this image has it (/opt/conda/lib: HDF5 1.10.6, found in round 5 -- rounds 1 - 4 believed the
image had none and kept an h5py test that skipped everywhere) and through ctypes.util; `load()` returns None when there is none.
The consumers of the snapshot files go through this library: plot/plot2d.py:30-54 (h5py) and src/io/read_write_hdf5.rs:38-188
(the hdf5 crate)."""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

hid_t = C.c_int64
herr_t = C.c_int
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL = 0, 0
_LIB = None


def load():
    """The library handle (H5open done, automatic error printing off) or None."""
    global _LIB
    if _LIB is not None:
        return _LIB or None
    names = [os.environ.get("RPDE_LIBHDF5", "")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + \
            [ctypes.util.find_library("hdf5") or "", "libhdf5.so", "libhdf5_serial.so"]
    for n in names:
        if not n:
            continue
        try:
            lib = C.CDLL(n)
        except OSError:
            continue
        if not hasattr(lib, "H5Fopen"):
            continue
        sig = {
            "H5open": (herr_t, []), "H5get_libversion": (herr_t, [C.POINTER(C.c_uint)] * 3),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
            "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
            "H5Fclose": (herr_t, [hid_t]),
            "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            "H5Gclose": (herr_t, [hid_t]), "H5Gget_num_objs": (herr_t, [hid_t, C.POINTER(hsize_t)]),
            "H5Gget_objname_by_idx": (C.c_ssize_t, [hid_t, hsize_t, C.c_char_p, C.c_size_t]),
            "H5Gget_objtype_by_idx": (C.c_int, [hid_t, hsize_t]),
            "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dclose": (herr_t, [hid_t]), "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Sclose": (herr_t, [hid_t]),
            "H5Tclose": (herr_t, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
        }
        try:
            for name, (res, args) in sig.items():
                f = getattr(lib, name)
                f.restype, f.argtypes = res, args
        except AttributeError:
            continue
        lib.H5open()
        lib.H5Eset_auto2(0, None, None)
        lib.path = n
        _LIB = lib
        return lib
    _LIB = False
    return None


def version():
    lib = load()
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    lib.H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return (a.value, b.value, c.value)


class File:
    """with File(name, "r" | "a" | "w") as f: f.datasets() / f.read(path) / f.write(path, array)"""

    def __init__(self, name, mode="r"):
        self.lib = load()
        assert self.lib is not None, "no libhdf5"
        b = name.encode()
        if mode == "w":
            self.id = self.lib.H5Fcreate(b, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)   # the library's defaults: earliest format,
        else:                                                                            # symbol-table groups, contiguous datasets
            self.id = self.lib.H5Fopen(b, H5F_ACC_RDONLY if mode == "r" else H5F_ACC_RDWR, H5P_DEFAULT)
        if self.id < 0:
            raise OSError(f"libhdf5 cannot open {name!r} (mode {mode})")
        self.f64 = hid_t.in_dll(self.lib, "H5T_NATIVE_DOUBLE_g").value

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.id >= 0:
            assert self.lib.H5Fclose(self.id) >= 0
            self.id = -1

    def _members(self, gid):
        n = hsize_t()
        assert self.lib.H5Gget_num_objs(gid, C.byref(n)) >= 0
        out = []
        for i in range(n.value):
            buf = C.create_string_buffer(256)
            assert self.lib.H5Gget_objname_by_idx(gid, i, buf, 256) > 0
            out.append((buf.value.decode(), self.lib.H5Gget_objtype_by_idx(gid, i)))   # 0 group, 1 dataset
        return out

    def paths(self):
        """every dataset of the file, as 'group/name' paths"""
        out = []

        def walk(gid, prefix):
            for name, kind in self._members(gid):
                if kind == 0:
                    sub = self.lib.H5Gopen2(gid, name.encode(), H5P_DEFAULT)
                    assert sub >= 0, name
                    walk(sub, prefix + name + "/")
                    self.lib.H5Gclose(sub)
                elif kind == 1:
                    out.append(prefix + name)
        root = self.lib.H5Gopen2(self.id, b"/", H5P_DEFAULT)
        walk(root, "")
        self.lib.H5Gclose(root)
        return sorted(out)

    def read(self, path):
        d = self.lib.H5Dopen2(self.id, path.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(path)
        t = self.lib.H5Dget_type(d)
        cls, size = self.lib.H5Tget_class(t), self.lib.H5Tget_size(t)
        self.lib.H5Tclose(t)
        assert (cls, size) in ((1, 8), (0, 8)), f"{path}: class {cls} size {size}, expected f64 or a 64-bit integer"   # H5T_FLOAT = 1, H5T_INTEGER = 0
        mem, dt = (self.f64, np.float64) if cls == 1 else (hid_t.in_dll(self.lib, "H5T_NATIVE_UINT64_g").value, np.uint64)
        s = self.lib.H5Dget_space(d)
        nd = self.lib.H5Sget_simple_extent_ndims(s)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            self.lib.H5Sget_simple_extent_dims(s, dims, None)
        self.lib.H5Sclose(s)
        a = np.empty(tuple(dims[i] for i in range(nd)), dtype=dt)
        assert self.lib.H5Dread(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) >= 0, path
        self.lib.H5Dclose(d)
        return a

    def datasets(self):
        return {p: self.read(p) for p in self.paths()}

    def write(self, path, array):
        """create the dataset (and its groups) with the library's defaults and write it"""
        a = np.ascontiguousarray(array, dtype=np.float64)
        parts = path.split("/")
        loc, opened = self.id, []
        for g in parts[:-1]:
            sub = self.lib.H5Gopen2(loc, g.encode(), H5P_DEFAULT)
            if sub < 0:
                sub = self.lib.H5Gcreate2(loc, g.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            assert sub >= 0, g
            opened.append(sub)
            loc = sub
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        s = self.lib.H5Screate_simple(a.ndim, dims, None)
        d = self.lib.H5Dcreate2(loc, parts[-1].encode(), self.f64, s, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        assert d >= 0, path
        assert self.lib.H5Dwrite(d, self.f64, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) >= 0, path
        self.lib.H5Dclose(d)
        self.lib.H5Sclose(s)
        for g in reversed(opened):
            self.lib.H5Gclose(g)
