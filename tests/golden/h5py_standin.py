"""
A stand-in for the part of ``h5py`` the reference's writer / loader uses (``sparseSpatialSampling/data.py``: ``File`` in
modes r / w / a, ``keys()``, ``in``, ``[...]``, ``get``, ``create_group``, ``create_dataset(name, data=...)``, ``[()]``,
``.shape``, ``.dtype``, ``close``, context manager), written with ctypes directly on the HDF5 C library of the image --
the library h5py itself wraps.  It shares NO code with the product's HDF5 layer (``sparsespatialsampling_amd/h5io.py`` /
``libs3h5.so``), so that files written by either side are judged by an independent binding.

TEST TOOLING ONLY, like the numba / flowtorch / shapely stand-ins of ``ref_stubs.py``: it lets the REAL reference
``ExportData.export`` / ``Datawriter`` / ``XDMFWriter`` / ``Dataloader`` run in the development container (h5py is not
installed and cannot be).  What it pins is the reference's naming, shape, dtype, ordering and batching logic -- not h5py:
the h5py behaviours it restates are
  * ``keys()`` / iteration in name order (h5py's default: name index, no creation-order tracking),
  * ``create_dataset`` / ``create_group`` create missing intermediate groups and raise ``ValueError`` when the name exists,
  * ``create_dataset(name, data=None)`` without shape / dtype raises ``TypeError``,
  * ``get(path)`` returns ``None`` for a missing path, ``[path]`` raises ``KeyError``,
  * ``data=`` goes through ``numpy.asarray`` (torch CPU tensors included) and keeps its dtype; ``dataset[()]`` of a 0-d
    dataset is a numpy scalar.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0x0, 0x1, 0x2, 0x4
H5P_DEFAULT, H5S_ALL = 0, 0
H5S_SCALAR = 0
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5I_GROUP, H5I_DATASET = 2, 5
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE = 0


def _find_library():
    names = [os.environ.get("S3_HDF5_LIB"), "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so",
             ctypes.util.find_library("hdf5"), ctypes.util.find_library("hdf5_serial")]
    for n in names:
        if n:
            try:
                return C.CDLL(n)
            except OSError:
                continue
    raise ImportError("h5py stand-in: no HDF5 C library found")


_L = _find_library()


def _fn(name, res, *args):
    f = getattr(_L, name)
    f.restype, f.argtypes = res, list(args)
    return f


H5open = _fn("H5open", herr_t)
H5Eset_auto2 = _fn("H5Eset_auto2", herr_t, hid_t, C.c_void_p, C.c_void_p)
H5Fcreate = _fn("H5Fcreate", hid_t, C.c_char_p, C.c_uint, hid_t, hid_t)
H5Fopen = _fn("H5Fopen", hid_t, C.c_char_p, C.c_uint, hid_t)
H5Fclose = _fn("H5Fclose", herr_t, hid_t)
H5Fflush = _fn("H5Fflush", herr_t, hid_t, C.c_int)
H5Gcreate2 = _fn("H5Gcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t)
H5Gopen2 = _fn("H5Gopen2", hid_t, hid_t, C.c_char_p, hid_t)
H5Gclose = _fn("H5Gclose", herr_t, hid_t)
H5Gget_num_objs = _fn("H5Gget_num_objs", herr_t, hid_t, C.POINTER(hsize_t))
H5Lexists = _fn("H5Lexists", C.c_int, hid_t, C.c_char_p, hid_t)
H5Lget_name_by_idx = _fn("H5Lget_name_by_idx", C.c_ssize_t, hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p,
                         C.c_size_t, hid_t)
H5Oopen = _fn("H5Oopen", hid_t, hid_t, C.c_char_p, hid_t)
H5Oclose = _fn("H5Oclose", herr_t, hid_t)
H5Iget_type = _fn("H5Iget_type", C.c_int, hid_t)
H5Pcreate = _fn("H5Pcreate", hid_t, hid_t)
H5Pclose = _fn("H5Pclose", herr_t, hid_t)
H5Pset_create_intermediate_group = _fn("H5Pset_create_intermediate_group", herr_t, hid_t, C.c_uint)
H5Screate = _fn("H5Screate", hid_t, C.c_int)
H5Screate_simple = _fn("H5Screate_simple", hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t))
H5Sclose = _fn("H5Sclose", herr_t, hid_t)
H5Sget_simple_extent_ndims = _fn("H5Sget_simple_extent_ndims", C.c_int, hid_t)
H5Sget_simple_extent_dims = _fn("H5Sget_simple_extent_dims", C.c_int, hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t))
H5Dcreate2 = _fn("H5Dcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
H5Dopen2 = _fn("H5Dopen2", hid_t, hid_t, C.c_char_p, hid_t)
H5Dclose = _fn("H5Dclose", herr_t, hid_t)
H5Dwrite = _fn("H5Dwrite", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
H5Dread = _fn("H5Dread", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
H5Dget_space = _fn("H5Dget_space", hid_t, hid_t)
H5Dget_type = _fn("H5Dget_type", hid_t, hid_t)
H5Tclose = _fn("H5Tclose", herr_t, hid_t)
H5Tget_class = _fn("H5Tget_class", C.c_int, hid_t)
H5Tget_size = _fn("H5Tget_size", C.c_size_t, hid_t)
H5Tget_sign = _fn("H5Tget_sign", C.c_int, hid_t)

H5open()
H5Eset_auto2(0, None, None)                      # no error stack on stderr: failures surface as Python exceptions here


def _glob(name):
    return hid_t.in_dll(_L, name).value


_NATIVE = {np.dtype(np.float32): _glob("H5T_NATIVE_FLOAT_g"), np.dtype(np.float64): _glob("H5T_NATIVE_DOUBLE_g"),
           np.dtype(np.int8): _glob("H5T_NATIVE_INT8_g"), np.dtype(np.uint8): _glob("H5T_NATIVE_UINT8_g"),
           np.dtype(np.int16): _glob("H5T_NATIVE_INT16_g"), np.dtype(np.uint16): _glob("H5T_NATIVE_UINT16_g"),
           np.dtype(np.int32): _glob("H5T_NATIVE_INT32_g"), np.dtype(np.uint32): _glob("H5T_NATIVE_UINT32_g"),
           np.dtype(np.int64): _glob("H5T_NATIVE_INT64_g"), np.dtype(np.uint64): _glob("H5T_NATIVE_UINT64_g")}
_LCPL = H5Pcreate(_glob("H5P_CLS_LINK_CREATE_ID_g"))
H5Pset_create_intermediate_group(_LCPL, 1)


def _b(path):
    return path.encode()


class _Node:
    """something with a file and an absolute path inside it"""

    def __init__(self, file, path):
        # (a File is its own root group: it must not hold a reference to itself -- the cycle would keep a dropped, still open File
        # alive until the garbage collector runs, and HDF5 refuses to reopen a file read-write while a read-only handle exists; h5py's
        # File objects close when their last reference goes, which data.py relies on for the XDMFWriter's handle)
        self._file_ref, self._path = (None if file is self else file), path

    @property
    def _file_obj(self):
        return self if self._file_ref is None else self._file_ref

    @property
    def _fid(self):
        fid = self._file_obj._id
        if not fid:
            raise ValueError("Invalid location identifier (file is closed)")
        return fid

    def _abs(self, name):
        if name.startswith("/"):
            return name
        return (self._path.rstrip("/") + "/" + name) if name not in ("", ".") else self._path

    @property
    def name(self):
        return self._path


class Dataset(_Node):
    def _open(self):
        d = H5Dopen2(self._fid, _b(self._path), H5P_DEFAULT)
        if d < 0:
            raise KeyError(f"Unable to open dataset {self._path!r}")
        return d

    def _describe(self):
        d = self._open()
        try:
            sp, tp = H5Dget_space(d), H5Dget_type(d)
            nd = H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                H5Sget_simple_extent_dims(sp, dims, None)
            cls, size, sign = H5Tget_class(tp), H5Tget_size(tp), H5Tget_sign(tp)
            H5Sclose(sp)
            H5Tclose(tp)
        finally:
            H5Dclose(d)
        if cls == H5T_FLOAT:
            dt = np.dtype(f"f{size}")
        elif cls == H5T_INTEGER:
            dt = np.dtype(f"{'u' if sign == H5T_SGN_NONE else 'i'}{size}")
        else:
            raise TypeError(f"h5py stand-in: unsupported type class {cls} of {self._path!r}")
        return tuple(int(dims[i]) for i in range(nd)), dt

    @property
    def shape(self):
        return self._describe()[0]

    @property
    def dtype(self):
        return self._describe()[1]

    def __getitem__(self, key):
        shape, dt = self._describe()
        out = np.empty(shape, dtype=dt)
        d = self._open()
        try:
            if H5Dread(d, _NATIVE[dt], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"Can't read data of {self._path!r}")
        finally:
            H5Dclose(d)
        return out[key]                               # [()] of a 0-d array is a numpy scalar, as with h5py


class _KeysView:
    def __init__(self, names):
        self._names = names

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, item):
        return item in self._names

    def __repr__(self):
        return f"<KeysViewHDF5 {self._names}>"


class Group(_Node):
    def _exists(self, path):
        """every link of the path exists (H5Lexists fails on a missing intermediate group)"""
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur += "/" + part
            if H5Lexists(self._fid, _b(cur), H5P_DEFAULT) <= 0:
                return False
        return True

    def _kind(self, path):
        if path == "/":
            return H5I_GROUP
        if not self._exists(path):
            return None
        o = H5Oopen(self._fid, _b(path), H5P_DEFAULT)
        if o < 0:
            return None
        kind = H5Iget_type(o)
        H5Oclose(o)
        return kind

    def _wrap(self, path):
        kind = self._kind(path)
        if kind == H5I_GROUP:
            return Group(self._file_obj, path)
        if kind == H5I_DATASET:
            return Dataset(self._file_obj, path)
        return None

    def get(self, name, default=None):
        obj = self._wrap(self._abs(name))
        return default if obj is None else obj

    def __getitem__(self, name):
        obj = self._wrap(self._abs(name))
        if obj is None:
            raise KeyError(f"Unable to open object (object '{name}' doesn't exist)")
        return obj

    def __contains__(self, name):
        return self._exists(self._abs(name))

    def keys(self):
        g = H5Gopen2(self._fid, _b(self._path), H5P_DEFAULT)
        if g < 0:
            raise KeyError(f"Unable to open group {self._path!r}")
        try:
            n = hsize_t(0)
            H5Gget_num_objs(g, C.byref(n))
            names = []
            for i in range(n.value):
                size = H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
                buf = C.create_string_buffer(size + 1)
                H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, size + 1, H5P_DEFAULT)
                names.append(buf.value.decode())
        finally:
            H5Gclose(g)
        return _KeysView(names)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def _writable(self):
        if self._file_obj.mode == "r":
            raise ValueError("Unable to create object (no write intent on file)")

    def create_group(self, name):
        self._writable()
        path = self._abs(name)
        if self._exists(path):
            raise ValueError(f"Unable to create group (name already exists): {path}")
        g = H5Gcreate2(self._fid, _b(path), _LCPL, H5P_DEFAULT, H5P_DEFAULT)
        if g < 0:
            raise ValueError(f"Unable to create group {path!r}")
        H5Gclose(g)
        return Group(self._file_obj, path)

    def create_dataset(self, name, shape=None, dtype=None, data=None):
        self._writable()
        if data is None:
            if shape is None and dtype is None:
                raise TypeError("One of data, shape or dtype must be specified")
            raise NotImplementedError("h5py stand-in: create_dataset without data")
        a = np.asarray(data, order="C", dtype=dtype)
        if a.dtype not in _NATIVE:
            raise TypeError(f"h5py stand-in: unsupported dtype {a.dtype}")
        if shape is not None and tuple(np.atleast_1d(shape)) != a.shape:
            a = a.reshape(shape)
        a = np.ascontiguousarray(a) if a.ndim else a
        path = self._abs(name)
        if self._exists(path):
            raise ValueError(f"Unable to create dataset (name already exists): {path}")
        if a.ndim == 0:
            sp = H5Screate(H5S_SCALAR)
        else:
            sp = H5Screate_simple(a.ndim, (hsize_t * a.ndim)(*a.shape), None)
        d = H5Dcreate2(self._fid, _b(path), _NATIVE[a.dtype], sp, _LCPL, H5P_DEFAULT, H5P_DEFAULT)
        if d < 0:
            H5Sclose(sp)
            raise ValueError(f"Unable to create dataset {path!r}")
        rc = H5Dwrite(d, _NATIVE[a.dtype], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) if a.size else 0
        H5Dclose(d)
        H5Sclose(sp)
        if rc < 0:
            raise OSError(f"Can't write data of {path!r}")
        return Dataset(self._file_obj, path)


class File(Group):
    def __init__(self, name, mode="r"):
        name = os.fspath(name)
        self.filename, self.mode = name, mode
        self._id = 0
        if mode == "r":
            fid = H5Fopen(_b(name), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "r+":
            fid = H5Fopen(_b(name), H5F_ACC_RDWR, H5P_DEFAULT)
        elif mode == "w":
            fid = H5Fcreate(_b(name), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode in ("w-", "x"):
            fid = H5Fcreate(_b(name), H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "a":
            fid = H5Fopen(_b(name), H5F_ACC_RDWR, H5P_DEFAULT) if os.path.exists(name) else -1
            if fid < 0:
                fid = H5Fcreate(_b(name), H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError(f"Invalid mode; must be one of r, r+, w, w-, x, a (got {mode!r})")
        if fid < 0:
            if mode in ("r", "r+") and not os.path.exists(name):
                raise FileNotFoundError(f"Unable to open file (unable to open file: name = '{name}', errno = 2)")
            raise OSError(f"Unable to open file {name!r} in mode {mode!r}")
        self._id = fid
        Group.__init__(self, self, "/")

    def close(self):
        if self._id:
            fid, self._id = self._id, 0
            H5Fclose(fid)

    def flush(self):
        if self._id:
            H5Fflush(self._id, 1)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __bool__(self):
        return bool(self._id)
