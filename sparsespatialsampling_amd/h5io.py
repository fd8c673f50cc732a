"""
HDF5 access of the export path: ``open_h5(path, mode)`` returns a file object with the handful of operations the writer,
the loader and the XDMF writer need.

* ``NativeH5File`` -- libs3h5.so (``csrc/h5sink.cpp``, HDF5 C library, no h5py): synchronous writes / reads / listings and
  an asynchronous batch writer that stores the datasets of a whole snapshot batch from a snapshot-major host buffer in a
  background thread (include/s3h5.h).
* ``H5pyFile`` -- the same interface on h5py, used when the native library is not available.

Dataset paths are relative to the root ("grid/faces", "data/0.1/p_center"): the reference's on-disk layout
(data.py:361-430, export.py:283-299).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
H5_SO = os.environ.get("S3_H5_SO") or os.path.join(_HERE, "libs3h5.so")             # (S3_H5_SO: the sanitizer build of the CPU test job)

_CODES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int32): 2, np.dtype(np.int64): 3, np.dtype(np.uint8): 4}
_TYPES = {v: k for k, v in _CODES.items()}
EEXIST = -17

H5_SIGNATURES = {
    "s3h5_last_error": (C.c_char_p, []),
    "s3h5_version": (C.c_int, [C.POINTER(C.c_uint)] * 3),
    "s3h5_open": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "s3h5_close": (C.c_int, [C.c_void_p]),
    "s3h5_write": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "s3h5_write_snapshots_async": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int64]),
    "s3h5_write_snapshots_async_when": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]),
    "s3h5_flush": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "s3h5_wait_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "s3h5_exists": (C.c_int, [C.c_void_p, C.c_char_p]),
    "s3h5_shape": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]),
    "s3h5_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64]),
    "s3h5_list": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
}

_lib = None


class H5Error(OSError):
    """an HDF5 operation failed"""


def native_lib():
    """libs3h5.so, or None when it (or the HDF5 C library behind it) cannot be loaded"""
    global _lib
    if _lib is None:
        try:
            lib = C.CDLL(H5_SO)
            for name, (res, args) in H5_SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
        except OSError:
            _lib = False
    return _lib or None


def _as_numpy(data):
    if hasattr(data, "detach"):                       # torch tensor
        data = data.detach().cpu().numpy()
    a = np.asarray(data)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    elif a.dtype not in _CODES:
        a = a.astype(np.float64 if a.dtype.kind == "f" else np.int64)
    return a if a.ndim == 0 else np.ascontiguousarray(a)        # (ascontiguousarray would turn a scalar into [1])


class NativeH5File:
    backend = "libs3h5"

    def __init__(self, path, mode="r"):
        self._lib = native_lib()
        if self._lib is None:
            raise H5Error("libs3h5.so is not available")
        self.path, self.mode = path, mode
        self._h = C.c_void_p(0)
        self._check(self._lib.s3h5_open(os.fsencode(path), mode.encode(), C.byref(self._h)), f"open {path!r}")
        self._keep = {}                                 # host buffers of queued writes, by address: the same two stages and one flag
                                                        # come back batch after batch -- the dict does not grow with the export

    def _check(self, rc, what):
        if rc < 0:
            msg = self._lib.s3h5_last_error().decode(errors="replace")
            if rc == -3:
                raise FileNotFoundError(f"{what}: {msg}")
            raise H5Error(f"{what}: {msg}")
        return rc

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._h = C.c_void_p(0)
            self._check(self._lib.s3h5_close(h), "close")
            self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown / error already reported
            pass

    # -- writing ---------------------------------------------------------------------------------------------------
    def write(self, path, data):
        """one dataset; False when it exists already (the reference logs and skips, data.py:404-407)"""
        a = _as_numpy(data)
        dims = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        rc = self._lib.s3h5_write(self._h, path.encode(), _CODES[a.dtype], a.ndim, dims, a.ctypes.data_as(C.c_void_p))
        if rc == EEXIST:
            return False
        self._check(rc, f"write {path!r}")
        return True

    def write_snapshots(self, times, name, host, group="data", ready=None):
        """``data/<times[i]>/<name>`` = ``host[i]`` for every i, queued and written in the background.  ``host`` is a
        contiguous snapshot-major array / CPU tensor ``[T, ...]``; it must not be modified before ``flush()`` or
        ``wait_buffer(host)``.  ``ready = (flag, value)``: the values are still being copied into ``host``; they are complete
        once the int32 host tensor ``flag[0] >= value`` (the producer writes it behind the copy) -- the writer waits for that."""
        t = host if hasattr(host, "data_ptr") else None
        a = host.numpy() if t is not None else np.ascontiguousarray(host)
        if not a.flags.c_contiguous or a.dtype not in _CODES or a.shape[0] != len(times):
            raise ValueError("write_snapshots: contiguous [T, ...] array of a supported type with one entry per write time required")
        names = (C.c_char_p * len(times))(*[str(s).encode() for s in times])
        dims = (C.c_int64 * max(a.ndim - 1, 1))(*a.shape[1:])
        stride = a.strides[0] if len(times) else 0
        flag, value = (C.c_void_p(ready[0].data_ptr()), int(ready[1])) if ready is not None else (None, 0)
        self._check(self._lib.s3h5_write_snapshots_async_when(self._h, group.encode(), names, len(times), name.encode(),
                                                              _CODES[a.dtype], a.ndim - 1, dims, a.ctypes.data_as(C.c_void_p),
                                                              stride, flag, value),
                    f"queue {group}/*/{name}")
        self._keep[a.ctypes.data] = host
        if ready is not None:
            self._keep[ready[0].data_ptr()] = ready[0]

    def flush(self):
        """wait for the queued writes; returns how many datasets were skipped because they existed"""
        skipped = C.c_int64(0)
        self._check(self._lib.s3h5_flush(self._h, C.byref(skipped)), "flush")
        self._keep = {}
        return skipped.value

    def wait_buffer(self, host):
        a = host.numpy() if hasattr(host, "data_ptr") else host
        self._check(self._lib.s3h5_wait_buffer(self._h, a.ctypes.data_as(C.c_void_p), a.nbytes), "wait_buffer")

    # -- reading ---------------------------------------------------------------------------------------------------
    def exists(self, path):
        return bool(self._check(self._lib.s3h5_exists(self._h, path.encode()), f"exists {path!r}"))

    def shape(self, path):
        dtype, ndim, dims = C.c_int(0), C.c_int(0), (C.c_int64 * 8)()
        self._check(self._lib.s3h5_shape(self._h, path.encode(), C.byref(dtype), C.byref(ndim), dims), f"shape {path!r}")
        return tuple(dims[i] for i in range(ndim.value))

    def read(self, path):
        dtype, ndim, dims = C.c_int(0), C.c_int(0), (C.c_int64 * 8)()
        self._check(self._lib.s3h5_shape(self._h, path.encode(), C.byref(dtype), C.byref(ndim), dims), f"shape {path!r}")
        shape = tuple(dims[i] for i in range(ndim.value))
        code = dtype.value if dtype.value in _TYPES else 1
        out = np.empty(shape, dtype=_TYPES[code])
        self._check(self._lib.s3h5_read(self._h, path.encode(), code, out.ctypes.data_as(C.c_void_p), out.size), f"read {path!r}")
        return out

    def keys(self, group="/"):
        need, n = C.c_size_t(0), C.c_int64(0)
        self._check(self._lib.s3h5_list(self._h, group.encode(), None, 0, C.byref(need), C.byref(n)), f"list {group!r}")
        buf = C.create_string_buffer(need.value)
        self._check(self._lib.s3h5_list(self._h, group.encode(), buf, need.value, C.byref(need), C.byref(n)), f"list {group!r}")
        return [s for s in buf.value.decode().split("\n") if s]


class H5pyFile:
    """the same interface on h5py (synchronous)"""
    backend = "h5py"

    def __init__(self, path, mode="r"):
        import h5py
        self.path, self.mode = path, mode
        self._f = h5py.File(path, mode)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None

    def write(self, path, data):
        if path in self._f:
            return False
        self._f.create_dataset(path, data=_as_numpy(data))
        return True

    def write_snapshots(self, times, name, host, group="data", ready=None):
        if ready is not None:                       # (synchronous backend: wait for the producer's word here)
            import time
            while int(ready[0][0]) < int(ready[1]):
                time.sleep(50e-6)
        a = host.numpy() if hasattr(host, "data_ptr") else np.asarray(host)
        self._skipped = getattr(self, "_skipped", 0)
        for i, t in enumerate(times):
            self._skipped += not self.write(f"{group}/{t}/{name}", a[i])

    def flush(self):
        n, self._skipped = getattr(self, "_skipped", 0), 0
        return n

    def wait_buffer(self, host):
        pass

    def exists(self, path):
        return path in self._f

    def shape(self, path):
        return tuple(self._f[path].shape)

    def read(self, path):
        return self._f[path][()]

    def keys(self, group="/"):
        return list(self._f[group].keys())


def open_h5(path, mode="r"):
    """the native sink when libs3h5.so loads, h5py otherwise"""
    if native_lib() is not None:
        return NativeH5File(path, mode)
    try:
        return H5pyFile(path, mode)
    except ImportError as err:
        raise ImportError("Neither libs3h5.so (build it with __graft_entry__.build(); needs the HDF5 C library) nor h5py is "
                          "available: the HDF5 / XDMF export cannot be written.") from err
