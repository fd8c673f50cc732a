"""
Thin Python front-end of the libs3hip.so C ABI (include/s3hip.h).

torch is used here only as plumbing: device memory (``torch.empty(..., device="cuda")``), host<->device copies and the
current HIP stream.  Every function below ends in exactly one hand-written HIP kernel family of libs3hip.so; no torch
operator runs on the hot path.  All tensors passed in must live on the current CUDA(HIP) device and be contiguous.
"""
import ctypes as C
import os

import numpy as np
import torch as pt

from . import _lib
from ._lib import check

DTYPE_CODE = {pt.float32: 0, pt.float64: 1}


def _stream():
    """torch's current HIP stream of the current device as a raw handle (the private getters skip the availability
    checks of ``torch.cuda.current_stream()``, which cost more than a small kernel launch)"""
    try:
        return C.c_void_p(pt._C._cuda_getCurrentRawStream(pt._C._cuda_getDevice()))
    except AttributeError:          # other torch build
        return C.c_void_p(pt.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    if not (t.is_cuda and t.is_contiguous()):       # a host pointer handed to a kernel would fault the GPU
        raise TypeError("device-resident contiguous tensor required")
    return C.c_void_p(t.data_ptr())


def _host_f64(values, n=3):
    a = np.zeros(n, dtype=np.float64)
    v = np.asarray(values, dtype=np.float64).ravel()
    a[:len(v)] = v
    return a


def device():
    """The torch device of the hot path; raises HipUnavailableError when there is none (no CPU fallback)."""
    _lib.require_device()
    if not pt.cuda.is_available():
        raise _lib.HipUnavailableError("torch sees no HIP device; it is needed for device memory and streams.")
    return pt.device("cuda", pt.cuda.current_device())


def synchronize():
    pt.cuda.current_stream().synchronize()


def to_device(x, dtype=None):
    """numpy / CPU tensor / CUDA tensor -> contiguous CUDA tensor (optionally cast).  Host tensors go through the native
    staged upload (s3_upload_rows: a memcpy into a page-locked buffer plus an asynchronous copy on the current stream -- a
    plain copy from pageable memory blocks the caller for ~0.2 ms however small it is, and the refine loop uploads two id
    lists per iteration), in 1-MiB rows when they are large."""
    if isinstance(x, np.ndarray):
        x = pt.from_numpy(np.ascontiguousarray(x))
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    dev = device()
    if x.is_cuda or not x.is_contiguous() or x.numel() == 0:
        return x.to(dev, non_blocking=False).contiguous()
    nbytes = x.numel() * x.element_size()
    if nbytes <= (4 << 20):
        out = pt.empty(x.shape, dtype=x.dtype, device=dev)
        rc = _lib.hip_lib().s3_upload_rows(C.c_void_p(x.data_ptr()), 1, nbytes, C.c_void_p(out.data_ptr()), nbytes, _stream())
        if rc == 0:
            return out                                          # (x was copied into the staging buffer: it may be reused)
        return x.to(dev, non_blocking=False).contiguous()       # no page-locked staging memory on this host
    # (r5: every size goes through the library's own page-locked staging buffers -- no pageable pointer is handed to the HIP
    # runtime, which would pin the caller's pages on the fly; round 5 saw rare GPU memory faults, "write access to a read-only page"
    # at a host heap address, in copies the runtime had pinned that way: HISTORY 9)
    out = pt.empty(x.shape, dtype=x.dtype, device=dev)
    flat_h, flat_d = x.reshape(-1), out.reshape(-1)
    row = (1 << 20) // x.element_size()                     # 1-MiB rows, pitch = row length
    n_full = flat_h.numel() // row
    try:
        upload_rows(flat_h[:n_full * row].view(n_full, row), flat_d[:n_full * row].view(n_full, row))
        tail = flat_h.numel() - n_full * row
        if tail:
            upload_rows(flat_h[n_full * row:].view(1, tail), flat_d[n_full * row:].view(1, tail))
    except _lib.S3HipError:                                 # no page-locked staging memory on this host: plain copy
        flat_d.copy_(flat_h)
    synchronize()                                           # the caller may release or overwrite x
    return out


def knn_occupancy(k, dim):
    """target points per bucket for a k-nearest search: the sphere that holds k neighbours should span about 1.5 bucket
    sides -- smaller buckets waste fewer visited points, larger ones need fewer rings.  Measured on MI355X (5*10^6
    uniform 3-D points, k = 26, 360 000 predictions, before the pruning of far buckets): occupancy 8 / 4 / 2 / 1.5 / 1 ->
    3.58 / 2.94 / 2.63 / 2.36 / 3.60 ms; with pruning 2 / 1.5 / 1 -> 2.03 / 2.01 / 2.91 ms"""
    return max(1.0, k / (13.0 if dim == 3 else 6.7))


class KnnIndex:
    """Bucket-grid KNN index over the original CFD points (replaces sklearn's kd-tree, s_cube.py:161-163,
    export.py:120,423)."""

    def __init__(self, points, target_occupancy=0.0):
        pts = to_device(points, pt.float64)
        assert pts.dim() == 2 and pts.shape[1] in (2, 3), "points must be [N, 2|3]"
        self.n, self.dim = int(pts.shape[0]), int(pts.shape[1])
        self._handle = C.c_void_p(0)
        check(_lib.hip_lib().s3_knn_create(_ptr(pts), self.n, self.dim, float(target_occupancy), _stream(),
                                           C.byref(self._handle)), "s3_knn_create")
        self._has_values = False
        nb, nr = C.c_int64(0), C.c_int64(0)
        check(_lib.hip_lib().s3_knn_info(self._handle, C.byref(nb), C.byref(nr)), "s3_knn_info")
        self.n_buckets, self.n_refined_buckets = nb.value, nr.value

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            _lib.hip_lib().s3_knn_destroy(self._handle)
            self._handle = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def set_values(self, y):
        y = to_device(y, pt.float64)
        assert y.numel() == self.n
        check(_lib.hip_lib().s3_knn_set_values(self._handle, _ptr(y), _stream()), "s3_knn_set_values")
        pt.cuda.current_stream().synchronize()      # y may be freed by the caller
        self._has_values = True

    def query(self, q, k):
        """-> (idx int32 [nq,k], dist f64 [nq,k]) on the device, ascending in (dist, idx)."""
        q = to_device(q, pt.float64)
        nq = int(q.shape[0])
        idx = pt.empty((nq, k), dtype=pt.int32, device=q.device)
        dist = pt.empty((nq, k), dtype=pt.float64, device=q.device)
        check(_lib.hip_lib().s3_knn_query(self._handle, _ptr(q), nq, int(k), _ptr(idx), _ptr(dist), _stream()),
              "s3_knn_query")
        return idx, dist

    def predict(self, q, k):
        q = to_device(q, pt.float64)
        nq = int(q.shape[0])
        out = pt.empty(nq, dtype=pt.float64, device=q.device)
        check(_lib.hip_lib().s3_idw_predict(self._handle, _ptr(q), nq, int(k), _ptr(out), _stream()), "s3_idw_predict")
        return out

    @property
    def handle(self):
        return self._handle


def idw_weights(dist):
    w = pt.empty_like(dist)
    check(_lib.hip_lib().s3_idw_weights(_ptr(dist), int(dist.shape[0]), int(dist.shape[1]), _ptr(w), _stream()),
          "s3_idw_weights")
    return w


def interp(w, idx, data, out=None):
    """out[c, ...] = sum_m w[c,m] * data[idx[c,m], ...]   (export.py:446-468).  w f64 [nc,k], idx int32 [nc,k],
    data f32/f64 [n_src, ...] -- all on the device; returns f64 [nc, ...] on the device."""
    if not (w.dtype == pt.float64 and idx.dtype == pt.int32 and data.dtype in DTYPE_CODE and w.shape == idx.shape):
        raise TypeError("interp: w must be float64 [nc,k], idx int32 [nc,k], data float32/float64")
    nc, k = int(w.shape[0]), int(w.shape[1])
    n_src = int(data.shape[0])
    row_len = int(np.prod(data.shape[1:])) if data.dim() > 1 else 1
    if out is None:
        out = pt.empty((nc,) + tuple(data.shape[1:]), dtype=pt.float64, device=data.device)
    check(_lib.hip_lib().s3_interp(_ptr(w), _ptr(idx), nc, k, _ptr(data), DTYPE_CODE[data.dtype], n_src, row_len,
                                   _ptr(out), _stream()), "s3_interp")
    return out


# ExportData queues its device-to-host copies on a stream of their own and lets them complete behind its back (the host-logic
# tests replace this module by CPU stand-ins that do not have the attribute: there every copy is immediate)
ASYNC_TRANSFERS = True


def yard_stream(src, dst, reads, writes, nontemporal=True):
    """yardstick (bench.py): hand-written streaming kernel, ``reads`` 16-byte vectors read per ``writes`` written
    (s3_yard_stream); -> (bytes read, bytes written)"""
    r, w = C.c_int64(0), C.c_int64(0)
    check(_lib.hip_lib().s3_yard_stream(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel() * src.element_size(),
                                        dst.numel() * dst.element_size(), int(reads), int(writes), int(bool(nontemporal)),
                                        _stream(), C.byref(r), C.byref(w)), "s3_yard_stream")
    return r.value, w.value


def snapshot_major(values, n_comp, n_snapshots, out=None):
    """device [nc, n_comp*T] (or [nc, n_comp, T]) f64 -> device [T, nc, n_comp]: contiguous snapshots for the HDF5 sink"""
    nc = int(values.shape[0])
    if out is None:
        out = pt.empty((int(n_snapshots), nc, int(n_comp)), dtype=pt.float64, device=values.device)
    elif not (out.is_cuda and out.is_contiguous() and out.dtype == pt.float64 and out.numel() == values.numel()):
        raise TypeError("snapshot_major: contiguous float64 device tensor of the batch's size required as out")
    check(_lib.hip_lib().s3_snapshot_major(_ptr(values), nc, int(n_comp), int(n_snapshots), _ptr(out), _stream()),
          "s3_snapshot_major")
    return out


def snapshot_major_rows(values, n_comp, n_snapshots, rows, n_out, out_ptr):
    """a shard's values, device [n_mine, n_comp*T] f64, -> rows ``rows`` (device int32 [n_mine]) of the snapshot-major batch
    buffer ``[T, n_out, n_comp]`` at device address ``out_ptr`` (host memory all ranks share, s3_snapshot_major_rows)"""
    if not (rows.is_cuda and rows.dtype == pt.int32 and rows.is_contiguous() and rows.numel() == values.shape[0]):
        raise TypeError("snapshot_major_rows: one int32 device row id per input row required")
    check(_lib.hip_lib().s3_snapshot_major_rows(_ptr(values), int(values.shape[0]), int(n_comp), int(n_snapshots), _ptr(rows),
                                                int(n_out), C.c_void_p(int(out_ptr)), _stream()), "s3_snapshot_major_rows")


_LAUNCH_SWITCHES = ("S3_PLAN_MIN_BLOCKS", "S3_STREAM_MIN_TILES", "S3_STREAM_MAX_CHUNKS", "S3_INPLACE_SHIFT", "S3_SHIFT_MIN_CHUNKS",
                    "S3_SHORT_STREAM", "S3_SHORT_NO_QUAD", "S3_SHORT_LDS_WEIGHTS", "S3_PLAN_SPLIT", "S3_PLAN_BRICK", "S3_PLAN_TAIL",
                    "S3_OUT_HOLD")
_switch_state = [None]


def reload_env():
    """the library parses its S3_* launch switches once; tell it to read them again (A/B tools, tests)"""
    check(_lib.hip_lib().s3_debug_reload_env(), "s3_debug_reload_env")
    _switch_state[0] = tuple(os.environ.get(k) for k in _LAUNCH_SWITCHES)


def _sync_switches():
    """a switch flipped through ``os.environ`` since the last planned launch reaches the library before the next one
    (``os.environ`` is the interpreter's own dict: reading it races with nothing)"""
    now = tuple(os.environ.get(k) for k in _LAUNCH_SWITCHES)
    if now != _switch_state[0]:
        reload_env()


class InterpPlan:
    """De-duplicated, LDS-tiled form of a static neighbour table (s3_interp_plan_*): build once per KNN cache, reuse for
    every snapshot batch.  ``centers`` (cell centres, [nc, dim]) gives the Hilbert-curve processing order."""

    def __init__(self, idx, n_src, centers=None, tile_cells=0):
        if not (idx.is_cuda and idx.dtype == pt.int32 and idx.is_contiguous() and idx.dim() == 2):
            raise TypeError("InterpPlan: idx must be a contiguous int32 [nc, k] device tensor")
        self.nc, self.k = int(idx.shape[0]), int(idx.shape[1])
        self.n_src = int(n_src)
        ctr = to_device(centers, pt.float64) if centers is not None else None
        dim = int(ctr.shape[1]) if ctr is not None else 0
        self._handle = C.c_void_p(0)
        check(_lib.hip_lib().s3_interp_plan_create(_ptr(idx), self.nc, self.k, self.n_src, _ptr(ctr), dim, int(tile_cells), _stream(),
                                                   C.byref(self._handle)), "s3_interp_plan_create")
        nt, nr = C.c_int64(0), C.c_int64(0)
        check(_lib.hip_lib().s3_interp_plan_info(self._handle, C.byref(nt), C.byref(nr)), "s3_interp_plan_info")
        self.n_tiles, self.total_rows = nt.value, nr.value
        self._w_ref = None              # the weights tensor whose values the plan holds (kept alive: see interp)
        self._w_version = None
        self.n_table = None             # rows of the full table behind the source rows (set_source_ids)

    def set_weights(self, w):
        """attach the weights of the table ([nc, k] float64, caller's cell order); the plan keeps a COPY in tile order.
        Call it again after rewriting ``w`` in place through a raw pointer (``idw_weights`` writes that way): such writes
        do not change the tensor's version counter, so ``interp`` cannot see them."""
        if not (w.is_cuda and w.dtype == pt.float64 and w.is_contiguous() and tuple(w.shape) == (self.nc, self.k)):
            raise TypeError("InterpPlan: weights must be a contiguous float64 [nc, k] device tensor")
        check(_lib.hip_lib().s3_interp_plan_set_weights(self._handle, _ptr(w), _stream()), "s3_interp_plan_set_weights")
        # the tensor itself, not its address: while the plan holds it the allocator cannot hand the same address to
        # another weights tensor, so `is` identifies it
        self._w_ref, self._w_version = w, w._version

    def set_source_ids(self, ids, n_table):
        """``ids`` (int32 device tensor [n_src]): the row of the caller's full table behind each source row the plan was built
        on; afterwards ``interp_src`` reads a full ``[n_table, ...]`` batch where it lies (s3_interp_plan_set_source_ids)"""
        if not (ids.is_cuda and ids.dtype == pt.int32 and ids.is_contiguous() and ids.numel() == self.n_src):
            raise TypeError("InterpPlan.set_source_ids: contiguous int32 device tensor with one id per source row required")
        check(_lib.hip_lib().s3_interp_plan_set_source_ids(self._handle, _ptr(ids), int(n_table), _stream()),
              "s3_interp_plan_set_source_ids")
        self.n_table = int(n_table)

    def partition(self, world):
        """cost-balanced leaf-cell shards (s3_interp_plan_partition): ``(order, cuts)`` -- ``order`` is the plan's processing
        order (device int32 [nc], position -> cell id: cells in Hilbert order) and rank r owns ``order[cuts[r]:cuts[r + 1]]``"""
        order = pt.empty(self.nc, dtype=pt.int32, device=device())
        cuts = (C.c_int64 * (int(world) + 1))()
        check(_lib.hip_lib().s3_interp_plan_partition(self._handle, int(world), _ptr(order), cuts, _stream()),
              "s3_interp_plan_partition")
        return order, [int(c) for c in cuts]

    def cost_profile(self, n_samples):
        """cumulative per-snapshot cost of the plan's tiles along its processing order at ``n_samples + 1`` equally spaced
        cell positions (s3_interp_plan_cost_profile) -> float64 numpy array"""
        out = np.zeros(int(n_samples) + 1, dtype=np.float64)
        check(_lib.hip_lib().s3_interp_plan_cost_profile(self._handle, int(n_samples), out.ctypes.data_as(C.c_void_p), _stream()),
              "s3_interp_plan_cost_profile")
        return out

    @staticmethod
    def _layout(data, k=None):
        """(row_len, in_stride) of a data matrix the planned kernels can read, None otherwise.  Every kernel takes rows
        that start on a 16-byte boundary and are readable up to the next multiple of 16 bytes (``padded_rows`` views always
        are; dense rows when their length is a multiple of 16 bytes).  Plans with the reference's neighbour counts
        (``k`` = 8 | 26) also take dense rows of any length >= 16 bytes where they lie (element alignment; the persistent
        kernel, s3hip.h)."""
        if data.dtype not in DTYPE_CODE or data.dim() < 1:
            return None
        epv = 16 // data.element_size()
        row_len = int(np.prod(data.shape[1:])) if data.dim() > 1 else 1
        if data.is_contiguous():
            in_stride = row_len
        elif data.dim() == 2 and data.stride(1) == 1 and data.stride(0) >= row_len:
            in_stride = int(data.stride(0))
        else:
            return None
        if in_stride >= 1 << 31:
            return None
        padded = (row_len + epv - 1) // epv * epv
        if in_stride % epv or in_stride < padded or (data.is_cuda and data.data_ptr() % 16):
            if not (k in (8, 26) and row_len >= epv):
                return None
        return row_len, in_stride

    @staticmethod
    def supports(k, data):
        """can a plan with ``k`` neighbours read ``data`` as it stands?  ``k = None``: by the 16-byte rule alone (every plan)"""
        return (k is None or k <= 64) and InterpPlan._layout(data, k) is not None

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            _lib.hip_lib().s3_interp_plan_destroy(self._handle)
            self._handle = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def interp(self, w, data, out=None):
        """``data`` [n_src, ...]: contiguous, or a column slice ``buf[:, :L]`` of a wider 2-D buffer (rows padded to a
        multiple of 128 bytes keep every staged segment on one cache line, see ``padded_rows``)."""
        if not (w.dtype == pt.float64 and tuple(w.shape) == (self.nc, self.k) and int(data.shape[0]) == self.n_src
                and data.dtype in DTYPE_CODE):
            raise TypeError("InterpPlan.interp: weights / data do not match the plan")
        row_len, in_stride, out = self._check_batch(data, out, "interp")
        if w is not self._w_ref or w._version != self._w_version:    # another tensor, or modified through torch: re-attach
            self.set_weights(w)
        _sync_switches()
        check(_lib.hip_lib().s3_interp_planned(self._handle, C.c_void_p(0), C.c_void_p(data.data_ptr()),
                                               DTYPE_CODE[data.dtype], row_len, in_stride, _ptr(out), _stream()),
              "s3_interp_planned")
        return out

    def _check_batch(self, data, out, who):
        layout = self._layout(data, self.k)
        if layout is None:
            raise TypeError(f"InterpPlan.{who}: source rows must be 16-byte aligned (see padded_rows), or dense rows of at "
                            f"least 16 bytes on a plan with k = 8 | 26")
        row_len, in_stride = layout
        if out is None:
            out = pt.empty((self.nc,) + tuple(data.shape[1:]), dtype=pt.float64, device=data.device)
        if not (data.is_cuda and out.is_cuda and out.is_contiguous() and out.dtype == pt.float64
                and out.numel() == self.nc * row_len):
            raise TypeError(f"InterpPlan.{who}: device tensors required, out must be contiguous float64 [nc, ...]")
        return row_len, in_stride, out

    def yard_loads(self, table, variant=0, in_place=True):
        """yardstick (bench.py): the loads of this plan on ``table`` and nothing else (s3_yard_plan_loads); -> staged bytes"""
        row_len, in_stride = self._layout(table, None)
        staged = C.c_int64(0)
        check(_lib.hip_lib().s3_yard_plan_loads(self._handle, C.c_void_p(table.data_ptr()), self.n_table if in_place else 0,
                                                row_len * table.element_size(), in_stride * table.element_size(), int(variant),
                                                _stream(), C.byref(staged)), "s3_yard_plan_loads")
        return staged.value

    def interp_src(self, table, out=None):
        """like ``interp`` for a FULL batch ``table`` [n_table, ...] that is read where it lies (no gather of the referenced
        rows first); needs ``set_weights`` and ``set_source_ids``"""
        if self.n_table is None or self._w_ref is None:
            raise RuntimeError("InterpPlan.interp_src: call set_weights and set_source_ids first")
        if int(table.shape[0]) != self.n_table or table.dtype not in DTYPE_CODE:
            raise TypeError("InterpPlan.interp_src: the table does not match the ids given to set_source_ids")
        row_len, in_stride, out = self._check_batch(table, out, "interp_src")
        _sync_switches()
        check(_lib.hip_lib().s3_interp_planned_src(self._handle, C.c_void_p(table.data_ptr()), DTYPE_CODE[table.dtype],
                                                   self.n_table, row_len, in_stride, _ptr(out), _stream()),
              "s3_interp_planned_src")
        return out


def to_host(t):
    """contiguous device tensor -> numpy array in pageable host memory through the native staged download (s3_download);
    the result is complete on return"""
    t = t.contiguous()
    out = np.empty(tuple(t.shape), dtype=pt.empty((), dtype=t.dtype).numpy().dtype)
    if out.nbytes:
        synchronize()                      # t was produced on torch's current stream or on an engine stream that was waited for
        check(_lib.hip_lib().s3_download(out.ctypes.data_as(C.c_void_p), C.c_void_p(t.data_ptr()), out.nbytes, None), "s3_download")
    return out


def upload_rows(host, rows):
    """contiguous host tensor [n_rows, row_len] -> device rows ``rows`` (a ``padded_rows`` view: the bytes between two
    rows are padding and may be overwritten) through the native staged upload (s3_upload_rows); asynchronous on the
    current stream"""
    if host.is_cuda or not host.is_contiguous() or host.dim() != 2 or host.dtype != rows.dtype:
        raise TypeError("upload_rows: contiguous 2-D host tensor of the rows' dtype required")
    if not rows.is_cuda or rows.dim() != 2 or rows.stride(1) != 1 or tuple(rows.shape) != tuple(host.shape):
        raise TypeError("upload_rows: 2-D device rows of the same shape with unit inner stride required")
    item = host.element_size()
    check(_lib.hip_lib().s3_upload_rows(C.c_void_p(host.data_ptr()), int(host.shape[0]), int(host.shape[1]) * item,
                                        C.c_void_p(rows.data_ptr()), int(rows.stride(0)) * item, _stream()),
          "s3_upload_rows")
    return rows


def upload_rows_indexed(host, row_ids, rows):
    """rows ``row_ids`` (ascending int32 numpy array) of the contiguous host tensor [n, row_len] -> device rows ``rows``
    [len(row_ids), row_len] (see ``upload_rows``)"""
    if host.is_cuda or not host.is_contiguous() or host.dim() != 2 or host.dtype != rows.dtype:
        raise TypeError("upload_rows_indexed: contiguous 2-D host tensor of the rows' dtype required")
    ids = np.ascontiguousarray(row_ids, dtype=np.int32)
    if not rows.is_cuda or rows.dim() != 2 or rows.stride(1) != 1 or tuple(rows.shape) != (len(ids), int(host.shape[1])):
        raise TypeError("upload_rows_indexed: device rows [len(row_ids), row_len] with unit inner stride required")
    if len(ids) and (int(ids.min()) < 0 or int(ids.max()) >= int(host.shape[0])):
        raise IndexError("upload_rows_indexed: row id outside the host tensor")
    item = host.element_size()
    check(_lib.hip_lib().s3_upload_rows_indexed(C.c_void_p(host.data_ptr()), ids.ctypes.data_as(C.c_void_p), len(ids),
                                                int(host.shape[1]) * item, C.c_void_p(rows.data_ptr()),
                                                int(rows.stride(0)) * item, _stream()), "s3_upload_rows_indexed")
    return rows


def upload_row_pieces(host, row_ids, t0, t1, rows):
    """snapshots ``[t0, t1)`` of the rows ``row_ids`` (int32 numpy array, or None: all rows) of the contiguous host field
    ``host`` [N, n_comp, T] -> device rows ``rows`` [n_sel, n_comp * (t1 - t0)] (unit inner stride; the bytes between two rows
    may be overwritten) through the native staged upload (s3_upload_row_pieces); asynchronous on the current stream"""
    if host.is_cuda or not host.is_contiguous() or host.dim() != 3 or host.dtype != rows.dtype:
        raise TypeError("upload_row_pieces: contiguous host tensor [N, n_comp, T] of the rows' dtype required")
    n, n_comp, t = (int(v) for v in host.shape)
    if not 0 <= t0 < t1 <= t:
        raise ValueError(f"upload_row_pieces: snapshots [{t0}, {t1}) of {t}")
    ids = None if row_ids is None else np.ascontiguousarray(row_ids, dtype=np.int32)
    n_sel = n if ids is None else len(ids)
    if not rows.is_cuda or rows.dim() != 2 or rows.stride(1) != 1 or tuple(rows.shape) != (n_sel, n_comp * (t1 - t0)):
        raise TypeError("upload_row_pieces: device rows [n_sel, n_comp * (t1 - t0)] with unit inner stride required")
    if ids is not None and len(ids) and (int(ids.min()) < 0 or int(ids.max()) >= n):
        raise IndexError("upload_row_pieces: row id outside the host tensor")
    item = host.element_size()
    check(_lib.hip_lib().s3_upload_row_pieces(C.c_void_p(host.data_ptr()), None if ids is None else ids.ctypes.data_as(C.c_void_p),
                                              n_sel, n_comp * t * item, t0 * item, n_comp, (t1 - t0) * item, t * item,
                                              C.c_void_p(rows.data_ptr()), int(rows.stride(0)) * item, _stream()),
          "s3_upload_row_pieces")
    return rows


def gather_rows(src, ids, dst):
    """device rows: ``dst[i, :] = src[ids[i], :]`` (``ids`` int32 device tensor) or ``dst[i, :] = src[i, :]`` (``ids``
    None: a re-pitch).  ``src`` / ``dst`` are 2-D device tensors (or ``padded_rows`` views) with unit inner stride."""
    if not (src.is_cuda and dst.is_cuda and src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
            and src.dtype == dst.dtype and src.shape[1] == dst.shape[1]):
        raise TypeError("gather_rows: 2-D device tensors of one dtype and row length with unit inner stride required")
    n = int(dst.shape[0])
    if ids is not None and not (ids.is_cuda and ids.dtype == pt.int32 and ids.is_contiguous() and ids.numel() == n):
        raise TypeError("gather_rows: ids must be a contiguous int32 device tensor with one entry per destination row")
    item = src.element_size()
    check(_lib.hip_lib().s3_gather_rows(C.c_void_p(src.data_ptr()), int(src.shape[0]), int(src.shape[1]) * item,
                                        int(src.stride(0)) * item, _ptr(ids), n, C.c_void_p(dst.data_ptr()),
                                        int(dst.stride(0)) * item, _stream()), "s3_gather_rows")
    return dst


def exclusive_scan(x):
    """exclusive prefix sum of a contiguous int32 / int64 device tensor (csrc/scan_sort.h) -> new tensor"""
    if not (x.is_cuda and x.is_contiguous() and x.dtype in (pt.int32, pt.int64) and x.dim() == 1):
        raise TypeError("exclusive_scan: contiguous 1-D int32 / int64 device tensor required")
    out = pt.empty_like(x)
    check(_lib.hip_lib().s3_exclusive_scan(_ptr(x), _ptr(out), int(x.numel()), x.element_size(), _stream()), "s3_exclusive_scan")
    return out


def sort_pairs(keys, vals, bits=64):
    """stable ascending sort of (key, value) pairs by the low ``bits`` bits of the int64 keys (read as unsigned), in place"""
    if not (keys.is_cuda and vals.is_cuda and keys.is_contiguous() and vals.is_contiguous() and keys.dtype == pt.int64
            and vals.dtype == pt.int32 and keys.numel() == vals.numel()):
        raise TypeError("sort_pairs: contiguous int64 keys and int32 values of one length on the device required")
    check(_lib.hip_lib().s3_sort_pairs(_ptr(keys), _ptr(vals), int(keys.numel()), int(bits), _stream()), "s3_sort_pairs")
    return keys, vals


def spatial_order(points):
    """Hilbert-curve order of device points [n, 2|3] float64 -> int32 permutation (position -> point) on the device"""
    pts = to_device(points, pt.float64)
    perm = pt.empty(int(pts.shape[0]), dtype=pt.int32, device=pts.device)
    check(_lib.hip_lib().s3_spatial_order(_ptr(pts), int(pts.shape[0]), int(pts.shape[1]), _ptr(perm), _stream()),
          "s3_spatial_order")
    return perm


def referenced_rows(tables, n_src, coords=None):
    """the source rows the neighbour tables (int32 device tensors) reference: (used ids int32 [n_used] device, remap int32
    [n_src] device: position among the used rows or -1) -- mark / scan / compact on the device.  Ascending ids, or, with
    the points' coordinates ``coords`` [n_src, dim], the Hilbert-curve order of the referenced points."""
    dev = tables[0].device
    remap = pt.zeros(int(n_src), dtype=pt.int32, device=dev)
    for t in tables:
        check(_lib.hip_lib().s3_mark_rows(_ptr(t), int(t.numel()), int(n_src), _ptr(remap), _stream()), "s3_mark_rows")
    used = pt.empty(int(n_src), dtype=pt.int32, device=dev)
    n_used = C.c_int64(0)
    check(_lib.hip_lib().s3_compact_rows(_ptr(remap), int(n_src), _ptr(used), C.byref(n_used), _stream()),
          "s3_compact_rows")
    used = used[:n_used.value]
    if coords is not None and n_used.value > 1:
        # keep the referenced rows in Hilbert order of their coordinates instead of the CFD mesh's numbering
        pts = pt.empty((n_used.value, int(coords.shape[1])), dtype=pt.float64, device=dev)
        gather_rows(to_device(coords, pt.float64), used.contiguous(), pts)
        ordered = pt.empty((n_used.value, 1), dtype=pt.int32, device=dev)
        gather_rows(used.contiguous().reshape(-1, 1), spatial_order(pts), ordered)
        used = ordered.reshape(-1)
        check(_lib.hip_lib().s3_positions_of(_ptr(used), n_used.value, _ptr(remap), int(n_src), _stream()), "s3_positions_of")
    return used, remap


def remap_indices(idx, remap):
    """``idx[i] = remap[idx[i]]`` in place (int32 device tensors)"""
    check(_lib.hip_lib().s3_remap_indices(_ptr(idx), int(idx.numel()), _ptr(remap), int(remap.numel()), _stream()),
          "s3_remap_indices")
    return idx


def padded_rows(n_rows, row_len, dtype, dev, extra_lines=0):
    """[n_rows, row_len] view of a device buffer whose row pitch is a whole number of 128-byte lines (upload target for
    snapshot batches: every 128-B segment the planned kernel stages then sits on exactly one cache line).  For long rows
    the line count is moved to the next value = 3 (mod 4): a tile reads the same column of ~400 scattered rows at a
    time, and with a pitch of 2^n or 2^n + 1 lines those addresses load the memory channels unevenly (MI355X, 4000-B
    rows, same process: 32 lines 3.69 ms, 33: 3.66, 34: 3.41*, 35: 3.37*/3.54, 37..47: 3.52-3.53, 64: 3.44*; * = a
    faster box of the pool)"""
    item = pt.empty((), dtype=dtype).element_size()
    if row_len * item <= 64 and not extra_lines:
        # short rows (16 snapshots of a scalar field): whole 64-byte sectors, two rows per line -- half the footprint
        pitch = 16 // item if row_len * item <= 16 else (32 // item if row_len * item <= 32 else 64 // item)
        return pt.empty((n_rows, pitch), dtype=dtype, device=dev)[:, :row_len]
    per_line = 128 // item
    lines = (row_len + per_line - 1) // per_line
    if lines >= 16:
        while lines % 4 != 3 or lines % 32 == 31:
            lines += 1
    pitch = (lines + int(extra_lines)) * per_line
    return pt.empty((n_rows, pitch), dtype=dtype, device=dev)[:, :row_len]


# ---- refine kernels on the device-resident cell arrays ----------------------------------------------------------
def make_children(center, level, parents, new_index, width):
    dim = int(center.shape[1])
    check(_lib.hip_lib().s3_make_children(_ptr(center), _ptr(level), _ptr(parents), int(parents.numel()),
                                          int(new_index), dim, float(width), _stream()), "s3_make_children")


def child_gain(knn, k, center, level, first, n, width, level_factor, gain0, metric, gain, scratch):
    dim = int(center.shape[1])
    check(_lib.hip_lib().s3_child_gain(knn.handle, int(k), _ptr(center), _ptr(level), int(first), int(n), dim,
                                       float(width), _ptr(level_factor), float(gain0), _ptr(metric), _ptr(gain),
                                       _ptr(scratch), _stream()), "s3_child_gain")


def child_gain_reuse(knn, k, center, level, first, n, width, level_factor, gain0, metric, gain, scratch, parents,
                     parents_offset, child_metric):
    """``child_gain`` that keeps every cell's child predictions in ``child_metric`` [cap, 2^d] and, with ``parents`` (the
    batch's ordered parent ids), takes a new cell's centre value from its parent's entry instead of searching again"""
    dim = int(center.shape[1])
    check(_lib.hip_lib().s3_child_gain_reuse(knn.handle, int(k), _ptr(center), _ptr(level), int(first), int(n), dim,
                                             float(width), _ptr(level_factor), float(gain0), _ptr(metric), _ptr(gain),
                                             _ptr(scratch), _ptr(parents), int(parents_offset), _ptr(child_metric), _stream()),
          "s3_child_gain_reuse")


def mask_box(center, level, cells, first, n, width, lo, hi, refine_mode, keep_inside, invalid):
    dim = int(center.shape[1])
    lo, hi = _host_f64(lo), _host_f64(hi)
    check(_lib.hip_lib().s3_mask_box(_ptr(center), _ptr(level), _ptr(cells), int(first), int(n), dim, float(width),
                                     lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), int(refine_mode),
                                     int(keep_inside), _ptr(invalid), _stream()), "s3_mask_box")


def mask_sphere(center, level, cells, first, n, width, pos, radius, refine_mode, keep_inside, invalid):
    dim = int(center.shape[1])
    pos = _host_f64(pos)
    check(_lib.hip_lib().s3_mask_sphere(_ptr(center), _ptr(level), _ptr(cells), int(first), int(n), dim, float(width),
                                        pos.ctypes.data_as(C.c_void_p), float(radius), int(refine_mode),
                                        int(keep_inside), _ptr(invalid), _stream()), "s3_mask_sphere")


def mask_cylinder(center, level, cells, first, n, width, p0, axis, norm, r0, r1, is_cone, refine_mode, keep_inside,
                  invalid):
    p0, axis = _host_f64(p0), _host_f64(axis)
    check(_lib.hip_lib().s3_mask_cylinder(_ptr(center), _ptr(level), _ptr(cells), int(first), int(n), float(width),
                                          p0.ctypes.data_as(C.c_void_p), axis.ctypes.data_as(C.c_void_p), float(norm),
                                          float(r0), float(r1), int(is_cone), int(refine_mode), int(keep_inside),
                                          _ptr(invalid), _stream()), "s3_mask_cylinder")


def mask_polygon(center, level, cells, first, n, width, poly_dev, refine_mode, keep_inside, invalid):
    check(_lib.hip_lib().s3_mask_polygon(_ptr(center), _ptr(level), _ptr(cells), int(first), int(n), float(width),
                                         _ptr(poly_dev), int(poly_dev.shape[0]), int(refine_mode), int(keep_inside),
                                         _ptr(invalid), _stream()), "s3_mask_polygon")


def mask_triangle(center, level, cells, first, n, width, points, refine_mode, keep_inside, invalid):
    pts = _host_f64(points, 6)
    check(_lib.hip_lib().s3_mask_triangle(_ptr(center), _ptr(level), _ptr(cells), int(first), int(n), float(width),
                                          pts.ctypes.data_as(C.c_void_p), int(refine_mode), int(keep_inside),
                                          _ptr(invalid), _stream()), "s3_mask_triangle")


def mask_prism(center, level, cells, first, n, width, origin, axis, norm, dims, triangle, refine_mode, keep_inside,
               invalid):
    origin, axis, tri = _host_f64(origin), _host_f64(axis), _host_f64(triangle, 6)
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    check(_lib.hip_lib().s3_mask_prism(_ptr(center), _ptr(level), _ptr(cells), int(first), int(n), float(width),
                                       origin.ctypes.data_as(C.c_void_p), axis.ctypes.data_as(C.c_void_p), float(norm),
                                       dims.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p),
                                       int(refine_mode), int(keep_inside), _ptr(invalid), _stream()), "s3_mask_prism")


def mask_tetrahedra(center, level, cells, first, n, width, positions, normals, refine_mode, keep_inside, invalid):
    """positions [n_tets, 4, 3], normals [n_tets, 3, 4] (inward; column p belongs to point p)"""
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    nrm = np.ascontiguousarray(normals, dtype=np.float64)
    n_tets = int(pos.shape[0])
    if pos.shape != (n_tets, 4, 3) or nrm.shape != (n_tets, 3, 4):
        raise ValueError("mask_tetrahedra: positions [n_tets,4,3] and normals [n_tets,3,4] expected")
    check(_lib.hip_lib().s3_mask_tetrahedra(_ptr(center), _ptr(level), _ptr(cells), int(first), int(n), float(width),
                                            pos.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p), n_tets,
                                            int(refine_mode), int(keep_inside), _ptr(invalid), _stream()),
          "s3_mask_tetrahedra")


def commit_batch(leaf, gain, parents, first, n_new, invalid):
    n_par = int(parents.numel()) if parents is not None else 0
    check(_lib.hip_lib().s3_commit_batch(_ptr(leaf), _ptr(gain), _ptr(parents), n_par, int(first), int(n_new),
                                         _ptr(invalid), _stream()), "s3_commit_batch")


def sumsq_leaf(metric, leaf, begin, end, out, scratch):
    check(_lib.hip_lib().s3_sumsq_leaf(_ptr(metric), _ptr(leaf), int(begin), int(end), _ptr(out), _ptr(scratch),
                                       _stream()), "s3_sumsq_leaf")


def sumsq_blocks(metric, leaf, n_cells, block_begin, block_end, partial):
    check(_lib.hip_lib().s3_sumsq_blocks(_ptr(metric), _ptr(leaf), int(n_cells), int(block_begin), int(block_end),
                                         _ptr(partial), _stream()), "s3_sumsq_blocks")


def sum_ordered(values, n, out):
    check(_lib.hip_lib().s3_sum_ordered(_ptr(values), int(n), _ptr(out), _stream()), "s3_sum_ordered")


def topn_leaf(gain, leaf, n_cells, n_top, scratch):
    """-> numpy int32 ids ordered like heapq.nlargest(n_top, leaves, key=(gain, -id))  (s_cube.py:601-602)."""
    out = np.empty(max(int(n_top), 1), dtype=np.int32)
    cnt = C.c_int64(0)
    check(_lib.hip_lib().s3_topn_leaf(_ptr(gain), _ptr(leaf), int(n_cells), int(n_top),
                                      out.ctypes.data_as(C.c_void_p), C.byref(cnt), _ptr(scratch), _stream()),
          "s3_topn_leaf")
    return out[:cnt.value]


def topn_scratch(n_cells, n_top, dev):
    nbytes = _lib.hip_lib().s3_topn_scratch_bytes(int(n_cells), int(n_top))
    return pt.empty((nbytes + 7) // 8, dtype=pt.float64, device=dev)
