"""
Sparse spatial sampling (S^3) grid generation for 2-D / 3-D CFD data -- MI355X-native drop-in for the reference's
``sparseSpatialSampling/s_cube.py`` (class ``SamplingTree``, reference lines 86-1692).

Division of labour:

* **HBM / HIP kernels** (``tree_backend.HipTreeBackend`` -> libs3hip.so): KNN index over the original points, child
  centres, inverse-distance metric prediction at every new cell centre and its 2^d candidate child centres, the gain,
  the geometry predicates, the captured-metric reduction and the top-N gain selection.
* **host, native** (``csrc/topology.cpp`` -> libs3topo.so): neighbour links, shared-node numbering, renumbering.
* **host, Python** (this file): the control flow of ``refine()``.  The ids of new cells follow the iteration order of the
  reference's Python sets (hard part 1): the same set operations are issued in the same sequence on
  ``IntSet`` objects (``csrc/pyset.cpp``: CPython's set of small ints restated natively, slot-for-slot equal to the
  interpreter's tables), a batch of ids per call, so the numbering is reproduced bit for bit without a Python object per
  cell.  Only the optional 2:1-balance mode (``max_delta_level=True``), which adds ids one by one between topology
  queries, still uses the interpreter's own sets.

The reference rejects nothing on the CPU; this implementation has no CPU compute path: constructing a
``SamplingTree`` without a HIP device raises ``HipUnavailableError``.
"""
import ctypes as C
import logging
import os
from time import time
from typing import Union

import numpy as np
import torch as pt

from . import _lib
from .intset import IntSet, RangeSet

logger = logging.getLogger(__name__)
logging.basicConfig(level=logging.INFO, format='[%(asctime)s] %(levelname)-8s %(message)s', datefmt='%Y-%m-%d %H:%M:%S',
                    force=True)

# the reference switches torch to float64 globally when its modules are imported (s_cube.py:19, export.py:23); user
# scripts written against it rely on that
pt.set_default_dtype(pt.float64)

# neighbour slots and child / node positions (reference s_cube.py:22-29)
NB = {
    "w": 0, "nw": 1, "n": 2, "ne": 3, "e": 4, "se": 5, "s": 6, "sw": 7,
    "wl": 8, "nwl": 9, "nl": 10, "nel": 11, "el": 12, "sel": 13, "sl": 14, "swl": 15, "cl": 16,
    "wu": 17, "nwu": 18, "nu": 19, "neu": 20, "eu": 21, "seu": 22, "su": 23, "swu": 24, "cu": 25
}
CH = {"swu": 0, "nwu": 1, "neu": 2, "seu": 3, "swl": 4, "nwl": 5, "nel": 6, "sel": 7}


def _make_backend(vertices, target, k):
    """The compute backend of the product: HIP kernels on MI355X, nothing else."""
    from .tree_backend import HipTreeBackend
    return HipTreeBackend(vertices, target, k)


def _ordered(ids):
    """the elements of an ``IntSet`` / ``set`` in iteration order, int64.  READ-ONLY for an ``IntSet``: the array is the set's
    cached order, shared by every caller until the set changes (``IntSet.to_array``) -- index with it, pass it on, but take a
    copy before sorting or assigning in place (numpy refuses to write to it)."""
    if isinstance(ids, (IntSet, RangeSet)):
        return ids.to_array()
    return np.fromiter(ids, dtype=np.int64, count=len(ids))


class _Topology:
    """numpy-facing wrapper of the native topology engine (csrc/topology.cpp)."""

    def __init__(self, dim, width, root_center):
        self._lib = _lib.topo_lib()
        self.dim, self.nch, self.nnb = dim, 2 ** dim, 8 if dim == 2 else 26
        rc = np.ascontiguousarray(root_center, dtype=np.float64)
        self._h = C.c_void_p(self._lib.s3t_create(dim, float(width), rc.ctypes.data_as(C.c_void_p)))
        if not self._h.value:
            raise RuntimeError("topology engine: could not be created (dimension must be 2 or 3)")
        # host-side shadow of what the refine loop itself needs (cell count, levels): the engine applies submitted
        # batches in its own thread and is only waited for when its tables are read
        self.n_created = 1
        self._level_shadow = np.zeros(4096, dtype=np.int32)

    # -- asynchronous updates (applied in submission order by the engine's worker thread) ----------------------------
    def _submit(self, kind, ids, relink=0):
        a = self._ids(ids)
        rc = self._lib.s3t_submit(self._h, kind, a.ctypes.data_as(C.c_void_p), len(a), int(relink))
        if rc != 0:
            raise MemoryError("topology engine: out of host memory")

    def submit_refine(self, parents, relink):
        """children of the ordered parents (ids known up front: consecutive from the current cell count); returns the id
        of the first new cell"""
        p = self._ids(parents)
        first, n_new = self.n_created, len(p) * self.nch
        if first + n_new > len(self._level_shadow):
            grown = np.zeros(max(2 * len(self._level_shadow), first + n_new), dtype=np.int32)
            grown[:first] = self._level_shadow[:first]
            self._level_shadow = grown
        self._level_shadow[first:first + n_new] = np.repeat(self._level_shadow[p] + 1, self.nch)
        self.n_created = first + n_new
        self._submit(0, p, relink)
        return first

    def submit_relink_parent_of(self, cells):
        self._submit(1, cells)

    def submit_mark_invalid(self, cells):
        self._submit(2, cells)

    def sync(self):
        """wait for the submitted updates; raises if one of them failed (nothing to wait for once the engine is closed)"""
        if getattr(self, "_h_raw", None) is None:
            return
        rc = self._lib.s3t_sync(self._h)
        if rc == -2:
            raise MemoryError("topology engine: out of host memory")
        if rc != 0:
            raise RuntimeError("topology engine: tried to refine a cell that is not a leaf")
        if self._lib.s3t_n_cells(self._h) != self.n_created:
            raise RuntimeError("host topology and device cell arrays disagree about the ids of the new cells")

    @property
    def level_now(self):
        """levels of all cells created so far, including batches the engine has not applied yet"""
        return self._level_shadow[:self.n_created]

    def close(self):
        h = getattr(self, "_h_raw", None)
        if h is not None and h.value:
            self._lib.s3t_destroy(h)
        self._h_raw = None

    __del__ = close

    @property
    def _h(self):
        h = getattr(self, "_h_raw", None)
        if h is None:
            raise RuntimeError("topology closed")
        return h

    @_h.setter
    def _h(self, value):
        self._h_raw = value

    @property
    def n_cells(self):
        return self._lib.s3t_n_cells(self._h)

    @property
    def n_nodes(self):
        return self._lib.s3t_n_nodes(self._h)

    def _view(self, fn, dtype, shape):
        ptr = fn(self._h)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    # views are valid until the next refine() call
    @property
    def level(self):
        return self._view(self._lib.s3t_level, np.int32, (self.n_cells,))

    @property
    def parent(self):
        return self._view(self._lib.s3t_parent, np.int32, (self.n_cells,))

    @property
    def first_child(self):
        return self._view(self._lib.s3t_first_child, np.int32, (self.n_cells,))

    @property
    def nb(self):
        return self._view(self._lib.s3t_nb, np.int32, (self.n_cells, self.nnb))

    @property
    def node_idx(self):
        return self._view(self._lib.s3t_node_idx, np.int64, (self.n_cells, self.nch))

    @property
    def center(self):
        return self._view(self._lib.s3t_center, np.float64, (self.n_cells, self.dim))

    @property
    def nodes(self):
        return self._view(self._lib.s3t_nodes, np.float64, (self.n_nodes, self.dim))

    @staticmethod
    def _ids(a):
        return np.ascontiguousarray(a, dtype=np.int64)

    def refine(self, parents, relink):
        """synchronous form of ``submit_refine``"""
        first = self.submit_refine(parents, relink)
        self.sync()
        return first

    def relink_parent_of(self, cells):
        c = self._ids(cells)
        self._lib.s3t_relink_parent_of(self._h, c.ctypes.data_as(C.c_void_p), len(c))

    def mark_invalid(self, cells):
        c = self._ids(cells)
        self._lib.s3t_mark_invalid(self._h, c.ctypes.data_as(C.c_void_p), len(c))

    def check_nb(self, cell):
        out = np.empty(self.nnb, dtype=np.int64)
        n = self._lib.s3t_check_nb(self._h, int(cell), out.ctypes.data_as(C.c_void_p))
        return out[:n].tolist()

    def finalize(self, dtype=np.int64):
        """renumbered grid: faces [n_leaf, 2^d] (``dtype`` int32 / int64), nodes [n_nodes, d]"""
        self.sync()
        n_nodes = C.c_int64(0)
        n_leaf = self._lib.s3t_finalize(self._h, C.byref(n_nodes))
        if n_leaf < 0:
            raise MemoryError("topology engine: out of host memory")
        faces = np.empty((n_leaf, self.nch), dtype=dtype)
        nodes = np.empty((n_nodes.value, self.dim), dtype=np.float64)
        self._lib.s3t_export_grid(self._h, faces.ctypes.data_as(C.c_void_p), int(np.dtype(dtype) == np.int32),
                                  nodes.ctypes.data_as(C.c_void_p))
        return faces, nodes

    def gather_cells(self, ids):
        """(centres [n, d] float64, levels [n] int64) of the listed cells"""
        a = self._ids(ids)
        centers, levels = np.empty((len(a), self.dim), dtype=np.float64), np.empty(len(a), dtype=np.int64)
        self._lib.s3t_gather_cells(self._h, a.ctypes.data_as(C.c_void_p), len(a), centers.ctypes.data_as(C.c_void_p),
                                   levels.ctypes.data_as(C.c_void_p))
        return centers, levels


class _DeviceTopology(_Topology):
    """The same interface on the device-resident engine (csrc/topo_dev.hip, ``s3_topo_*``): the tables live in HBM, every
    update is a handful of kernel launches on the engine's own stream (it overlaps the KNN kernels of the refine loop the
    way the host engine's worker thread does), and the host only ever holds what it asks for -- table views are
    downloads, cached until the next update.  Used whenever the tree runs on the HIP backend without the 2:1-balance
    mode (``_check_nb`` reads single rows between updates, which the host engine serves better)."""

    _TABLES = {"level": (0, np.int32), "parent": (1, np.int32), "first_child": (2, np.int32), "nb": (3, np.int32),
               "node_idx": (4, np.int64), "center": (5, np.float64), "nodes": (6, np.float64)}

    def __init__(self, dim, width, root_center):
        from . import hipops
        self._ops = hipops
        hipops.device()
        self._hip = _lib.hip_lib()
        self.dim, self.nch, self.nnb = dim, 2 ** dim, 8 if dim == 2 else 26
        rc = np.ascontiguousarray(root_center, dtype=np.float64)
        h = C.c_void_p(0)
        hipops.check(self._hip.s3_topo_create(dim, float(width), rc.ctypes.data_as(C.c_void_p), C.byref(h)), "s3_topo_create")
        self._h = h
        self.n_created = 1
        self._level_shadow = np.zeros(4096, dtype=np.int32)
        self._n_nodes = self.nch
        self._cache = {}
        self._dirty = False

    def _submit(self, kind, ids, relink=0):
        a = self._ids(ids)
        ptr = a.ctypes.data_as(C.c_void_p)
        if kind == 0:
            rc = self._hip.s3_topo_refine(self._h, ptr, len(a), int(relink), None)
        elif kind == 1:
            rc = self._hip.s3_topo_relink_parent_of(self._h, ptr, len(a))
        else:
            rc = self._hip.s3_topo_mark_invalid(self._h, ptr, len(a))
        self._ops.check(rc, "s3_topo update")
        self._dirty = True
        self._cache = {}

    def sync(self):
        if not self._dirty:
            return
        n_cells, n_nodes, err = C.c_int64(0), C.c_int64(0), C.c_int(0)
        self._ops.check(self._hip.s3_topo_sync(self._h, C.byref(n_cells), C.byref(n_nodes), C.byref(err)), "s3_topo_sync")
        if err.value == 2:
            raise RuntimeError("topology engine: a node reference chain did not resolve (internal error)")
        if err.value != 0:
            raise RuntimeError("topology engine: tried to refine a cell that is not a leaf")
        if n_cells.value != self.n_created:
            raise RuntimeError("device topology and device cell arrays disagree about the ids of the new cells")
        self._n_nodes = n_nodes.value
        self._dirty = False

    def close(self):
        h = getattr(self, "_h_raw", None)
        if h is not None and h.value:
            self._hip.s3_topo_destroy(h)
        self._h_raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    @property
    def n_cells(self):
        self.sync()
        return self.n_created

    @property
    def n_nodes(self):
        self.sync()
        return self._n_nodes

    def _table(self, name):
        """host copy of a device table (downloaded once per state of the engine)"""
        self.sync()
        if name not in self._cache:
            which, dtype = self._TABLES[name]
            rows = self._n_nodes if name == "nodes" else self.n_created
            per = {"nb": self.nnb, "node_idx": self.nch, "center": self.dim, "nodes": self.dim}.get(name, 1)
            out = np.empty((rows, per) if per > 1 else (rows,), dtype=dtype)
            ptr = C.c_void_p(0)
            self._ops.check(self._hip.s3_topo_table(self._h, which, C.byref(ptr)), "s3_topo_table")
            # (staged through the library's page-locked lanes: the finished grid of C4 is ~1 GB, and a plain copy into fresh pageable
            # memory is the pattern the runtime serves by pinning the destination on the fly -- ADVICE r5)
            self._ops.check(self._hip.s3_download(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes, None), "s3_download")
            self._cache[name] = out
        return self._cache[name]

    level = property(lambda self: self._table("level"))
    parent = property(lambda self: self._table("parent"))
    first_child = property(lambda self: self._table("first_child"))
    nb = property(lambda self: self._table("nb"))
    node_idx = property(lambda self: self._table("node_idx"))
    center = property(lambda self: self._table("center"))
    nodes = property(lambda self: self._table("nodes"))

    def relink_parent_of(self, cells):
        self._submit(1, cells)

    def mark_invalid(self, cells):
        self._submit(2, cells)

    def check_nb(self, cell):
        nb, fc, level = self.nb[int(cell)], self.first_child, self.level
        return [int(q) for q in nb if q >= 0 and fc[q] == -1 and level[q] < level[int(cell)]]

    def finalize(self, dtype=np.int64):
        self.sync()
        n_leaf, n_nodes = C.c_int64(0), C.c_int64(0)
        self._ops.check(self._hip.s3_topo_finalize(self._h, C.byref(n_leaf), C.byref(n_nodes)), "s3_topo_finalize")
        dev = self._ops.device()
        faces = pt.empty((n_leaf.value, self.nch), dtype=pt.int32 if np.dtype(dtype) == np.int32 else pt.int64, device=dev)
        nodes = pt.empty((n_nodes.value, self.dim), dtype=pt.float64, device=dev)
        self._ops.check(self._hip.s3_topo_export_grid(self._h, C.c_void_p(faces.data_ptr()), int(np.dtype(dtype) == np.int32),
                                                      C.c_void_p(nodes.data_ptr())), "s3_topo_export_grid")
        return self._ops.to_host(faces), self._ops.to_host(nodes)

    def gather_cells(self, ids):
        a = self._ids(ids)
        dev = self._ops.device()
        centers = pt.empty((len(a), self.dim), dtype=pt.float64, device=dev)
        levels = pt.empty(len(a), dtype=pt.int64, device=dev)
        self._ops.check(self._hip.s3_topo_gather_cells(self._h, a.ctypes.data_as(C.c_void_p), len(a), C.c_void_p(centers.data_ptr()),
                                                       C.c_void_p(levels.data_ptr())), "s3_topo_gather_cells")
        return self._ops.to_host(centers), self._ops.to_host(levels)


def _make_topology(dim, width, root_center, backend, max_delta_level):
    """device-resident engine for trees on the HIP backend (S3_TOPOLOGY=host keeps the host engine), host engine for the
    2:1-balance mode and for the CPU test backend"""
    import os
    on_device = type(backend).__name__ == "HipTreeBackend" and not max_delta_level
    if on_device and os.environ.get("S3_TOPOLOGY", "device") != "host":
        return _DeviceTopology(dim, width, root_center)
    return _Topology(dim, width, root_center)


class Cell(object):
    """Read-only view of one cell with the attribute names of the reference's ``Cell`` (s_cube.py:32-83).  The tree
    itself is stored as arrays; views are created on demand (``tree._cells[i]``)."""

    def __init__(self, tree, index):
        self._tree = tree
        self.index = int(index)

    @property
    def level(self):
        return int(self._tree._topo.level[self.index])

    @property
    def center(self):
        return pt.from_numpy(self._tree._topo.center[self.index].copy())

    @property
    def parent(self):
        p = int(self._tree._topo.parent[self.index])
        return None if p < 0 else Cell(self._tree, p)

    @property
    def children(self):
        fc = int(self._tree._topo.first_child[self.index])
        if fc == -1:
            return None
        if fc == -2:
            return []
        return tuple(Cell(self._tree, fc + c) for c in range(self._tree._topo.nch))

    @property
    def nb(self):
        return [None if n < 0 else Cell(self._tree, n) for n in self._tree._topo.nb[self.index].tolist()]

    @property
    def node_idx(self):
        return self._tree._topo.node_idx[self.index].tolist()

    @property
    def metric(self):
        return self._tree._cell_values()["metric"][self.index]

    @property
    def gain(self):
        return self._tree._cell_values()["gain"][self.index]

    def leaf_cell(self) -> bool:
        return self.children is None


class _CellList:
    def __init__(self, tree):
        self._tree = tree

    def __len__(self):
        return self._tree._topo.n_cells

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [Cell(self._tree, j) for j in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return Cell(self._tree, i)

    def __iter__(self):
        return (Cell(self._tree, i) for i in range(len(self)))


class SamplingTree(object):
    def __init__(self, vertices: pt.Tensor, target: pt.Tensor, geometry_obj: list, n_cells: int = None,
                 uniform_level: int = 5, min_metric: float = 0.75, max_delta_level: bool = False,
                 n_cells_iter_start: int = None, n_cells_iter_end: int = None, n_jobs: int = 1,
                 relTol: Union[int, float] = 1e-3, reach_at_least: float = 0.75, pre_select: bool = False):
        """Same arguments as the reference (s_cube.py:87-132).  ``n_jobs`` is accepted for compatibility; the work
        runs on the GPU."""
        from multiprocessing import cpu_count
        self._pre_select = pre_select
        self._n_jobs = n_jobs if n_jobs is not None else cpu_count()
        self._max_delta_level = max_delta_level
        self._geometry = geometry_obj
        self._n_cells = 0
        self._min_metric = min_metric
        self._n_cells_max = n_cells
        self._min_level = uniform_level
        self._current_min_level = 0
        self._current_max_level = 0
        # cells refined per iteration: 0.1 % of the original grid (s_cube.py:147-154)
        n_orig = vertices.size(0)
        self._cells_per_iter_start = int(0.001 * n_orig) if n_cells_iter_start is None else n_cells_iter_start
        if self._cells_per_iter_start <= 0:
            self._cells_per_iter_start = 1
        self._cells_per_iter_end = self._cells_per_iter_start if n_cells_iter_end is None else n_cells_iter_end
        self._cells_per_iter = self._cells_per_iter_start
        self._cells_per_iter_last = 1e9
        self._reach_at_least = reach_at_least
        self._width = None
        self._n_dimensions = vertices.size(-1)
        self._k = 8 if self._n_dimensions == 2 else 26
        # IntSet = CPython's set restated natively (same iteration order); the 2:1-balance mode interleaves single
        # insertions with topology queries and stays on the interpreter's sets
        self._new_set = set if max_delta_level else IntSet
        # (S3_LEAF_SET_THREAD=0: every update of the leaf set on the caller's thread)
        self._leaf_cells = set() if max_delta_level else IntSet(deferred=os.environ.get("S3_LEAF_SET_THREAD", "1") != "0")
        self._n_cells_after_uniform = None
        self._N_cells_per_iter = []
        self._final_nodes = None
        self.all_centers = []
        self.all_levels = None
        self.face_ids = None
        self._metric = []
        self._n_cells_log = []
        self._n_cells_orig = target.size(0)
        self.data_final_mesh = {}
        self._times = _initialize_time_dict()
        if relTol is None:
            self._relTol = 1e-3 if n_cells is None else 10
        else:
            self._relTol = relTol
        self._print_settings()

        # KNN index + metric go to the device (replaces KNeighborsRegressor.fit, s_cube.py:161-163)
        self._backend = _make_backend(vertices.detach().cpu().to(pt.float64).numpy(),
                                      target.detach().cpu().to(pt.float64).numpy(), self._k)
        self._values = None
        self._topo_engine = None
        self._cells = _CellList(self)
        self._create_first_cell()
        # like the reference (s_cube.py:205) the norm is taken in the dtype the user passed (a float32 metric gives a
        # float32 norm)
        self._target_norm = pt.linalg.norm(target.detach().cpu()).item()

    # ------------------------------------------------------------------------------------------------------------
    def _create_first_cell(self) -> None:
        """root cell from the ``keep_inside`` geometry (s_cube.py:338-397)"""
        middle_ = None
        for g in self._geometry:
            if g.keep_inside:
                self._width = g.main_width
                middle_ = g.center
            if g.center.size(0) != self._n_dimensions:
                raise ValueError(f"The number of dimensions for geometry object '{g.name}' with dim = {g.center.size(0)} "
                                 f"is not matching the number of dimensions within the CFD grid with dim = "
                                 f"{self._n_dimensions}.")
        if middle_ is None:
            raise ValueError("No GeometryObject with 'keep_inside=True', representing the numerical domain, was found.")

        nd = self._n_dimensions
        dirs = _directions(nd)
        root = middle_.detach().cpu().type(pt.float64).numpy()
        queries = np.repeat(root[None, :], 2 ** nd + 1, axis=0)
        queries[1:, :] += dirs * 0.25 * self._width
        metric = self._backend.predict(queries)

        # gain of the root: (width/2)^d * sum |m0 - mi|, accumulated like Python's sum() (s_cube.py:375-381)
        sum_distances = sum([abs(metric[0] - metric[i]) for i in range(1, len(metric))])
        gain = pow(self._width / 2, nd) * sum_distances
        if abs(gain - 0) < 1e-6:
            gain = 1.0
        self._gain0 = float(gain)
        self._n_cells += 1
        self._topo_engine = _make_topology(nd, self._width, root, self._backend, self._max_delta_level)
        self._backend.start(root, self._width, self._gain0, metric[0], self._gain0)
        self._leaf_cells.add(0)

    @property
    def _topo(self):
        """the native topology tables, up to date: waits for the batches the engine has not applied yet"""
        engine = self._topo_engine
        if engine is not None:
            engine.sync()
        return engine

    def _cell_values(self):
        """metric / gain of all cells on the host (lazy download; used by the ``Cell`` views and the parity tests)"""
        if self._values is None or len(self._values["metric"]) != self._topo_engine.n_created:
            self._values = self._backend.download(self._topo_engine.n_created)
        return self._values

    def _update_leaf_cells(self, idx_parents, idx_children) -> None:
        """``_leaf_cells -= all_parents; _leaf_cells.update(all_children)`` (s_cube.py:552-553, 897-898).  With the native
        sets the parents arrive as the ordered id array (discards do not depend on their order: the set ``all_parents`` is
        never built) and the children as a ``RangeSet``"""
        if isinstance(idx_parents, np.ndarray):
            self._leaf_cells.difference_update_ids(idx_parents)
        else:
            self._leaf_cells -= idx_parents
        self._leaf_cells.update(idx_children)

    def _batch_sets(self, order, first, n_new):
        """(all_parents, all_children) of a refine batch: Python sets in the 2:1-balance mode, otherwise the parents' id array
        and the virtual set of the consecutive new ids"""
        if self._new_set is set:
            all_parents, all_children = set(), set()
            all_parents.update(order.tolist())
            all_children.update(range(first, first + n_new))
            return all_parents, all_children
        return order, RangeSet(first, first + n_new)

    def _update_min_ref_level(self) -> None:
        level = self._topo_engine.level_now
        leaves = _ordered(self._leaf_cells)
        self._current_min_level = max(self._current_min_level, int(level[leaves].min()))

    def _check_stopping_criteria(self) -> bool:
        """True = keep refining (s_cube.py:263-284)"""
        if self._n_cells_max is None:
            if len(self._metric) > 1 and self._metric[-1] / self._min_metric >= self._reach_at_least:
                return self._metric[-1] < self._min_metric and abs(self._metric[-1] - self._metric[-2]) > self._relTol
        else:
            if len(self._leaf_cells) / self._n_cells_max >= self._reach_at_least:
                _relStop = abs(self._cells_per_iter / self._n_cells_max - self._cells_per_iter_last / self._n_cells_max)
                return len(self._leaf_cells) < self._n_cells_max and _relStop > self._relTol
        return True

    def _compute_n_cells_per_iter(self) -> None:
        """linear ramp of the batch size between start and end value (s_cube.py:286-315)"""
        if self._n_cells_max is None:
            _delta_x = self._min_metric - self._metric[0]
            _current_x = self._metric[-1]
        else:
            _delta_x = self._n_cells_max - self._n_cells_after_uniform
            _current_x = self._n_cells
        _delta_y = self._cells_per_iter_start - self._cells_per_iter_end
        _new = self._cells_per_iter_start - (_delta_y / _delta_x) * _current_x
        self._cells_per_iter_last = self._cells_per_iter
        self._cells_per_iter = int(_new) if _new > 1 else 1

    def _compute_captured_metric(self) -> bool:
        """||metric at the leaf centres||_2 / ||target||_2 (s_cube.py:317-336).  Each leaf's prediction was stored
        when the cell was created (the reference recomputes the identical values); the sum of squares is one device
        reduction over fixed 1024-cell blocks (multi-GPU: every rank a share of the blocks, one small all-gather; the result
        does not depend on the number of ranks, parallel.py)."""
        sumsq = self._backend.sumsq(self._topo_engine.n_created)
        _ratio = float(np.sqrt(sumsq)) / self._target_norm
        self._metric.append(_ratio)
        return _ratio < self._min_metric

    # ------------------------------------------------------------------------------------------------------------
    def _refine_batch(self, order: np.ndarray, uniform: bool):
        """create the children of the ordered parents: topology on the host, geometry + metric + gain on the device
        (body shared by s_cube.py:531-555 and 879-900).  Returns (first new id, number of new cells).  The ids of the new
        cells are known up front (children are numbered consecutively from the current cell count): the kernels are
        launched, the batch is handed to the topology engine's worker thread, and the caller carries on -- nothing the
        refine loop decides depends on the links or node ids (only the 2:1-balance mode reads them, through
        ``self._topo``, which waits)."""
        nch = 2 ** self._n_dimensions
        engine = self._topo_engine
        first = engine.n_created
        n_new = self._backend.refine_batch(order, first)
        if n_new != len(order) * nch:
            raise RuntimeError("host topology and device cell arrays disagree about the ids of the new cells")
        if engine.submit_refine(order, relink=uniform) != first:
            raise RuntimeError("host topology and device cell arrays disagree about the ids of the new cells")
        self._n_cells += n_new
        self._values = None
        return first, n_new

    def _refine_uniform(self) -> None:
        """level-synchronous refinement of all leaves (s_cube.py:508-561)"""
        logger.info("Starting uniform refinement.")
        self._times["t_start_uniform"] = time()
        for j in range(self._min_level):
            logger.info(f"\r\tStarting iteration no. {j}, N_cells = {len(self._leaf_cells)}")
            order = _ordered(self._leaf_cells)
            first, n_new = self._refine_batch(order, uniform=True)
            all_parents, all_children = self._batch_sets(order, first, n_new)
            self._update_leaf_cells(all_parents, all_children)
            self._current_min_level += 1
            self._current_max_level += 1
            # set(range(first, first + n_new)) of the reference == all_children (same insertions into an empty set, and
            # merging it into the leaf set did not touch it): its iteration order is reused instead of rebuilt
            self._remove_invalid_cells(all_children, _batch=(first, n_new))
        logger.info("Finished uniform refinement.")
        self._times["t_end_uniform"] = time()

    def _refine_cells(self, to_refine: set):
        """s_cube.py:865-902; returns the id range of the new cells"""
        order = _ordered(to_refine)
        if len(order):
            self._current_max_level = max(self._current_max_level, int(self._topo_engine.level_now[order].max()) + 1)
        if len(order) == 1 and not getattr(self, "_warned_single_cell", False):
            # deviation on an input the reference rejects: with exactly one cell to refine its `_compute_cell_centers` squeezes the
            # cell axis away (s_cube.py:445) and `_refine_cells` then indexes a 2-D tensor with three indices (s_cube.py:883)
            logger.warning("Exactly one cell is refined in this step: the reference implementation raises an IndexError here "
                           "(s_cube.py:445, 883); this implementation refines the cell and continues.")
            self._warned_single_cell = True
        first, n_new = self._refine_batch(order, uniform=False)
        all_parents, all_children = self._batch_sets(order, first, n_new)
        self._update_leaf_cells(all_parents, all_children)
        self._new_children = all_children          # == set(range(first, first + n_new)), reused by the callers
        return first, n_new

    def _remove_invalid_cells(self, _refined_cells: set, _refine_geometry: bool = False,
                              _geometry_no: Union[int, list] = None, _batch=None) -> Union[None, set]:
        """geometry verdict for the listed cells (s_cube.py:669-732).  ``_batch=(first, n)`` marks the call that
        follows a refine batch: the flags then also finish the batch's bookkeeping on the device."""
        if type(_geometry_no) is int:
            _geometry_no = [_geometry_no]
        _geometries = [self._geometry[g] for g in _geometry_no] if _geometry_no is not None else self._geometry

        order = _ordered(_refined_cells)
        if self._pre_select:
            # reference quirk (s_cube.py:1832-1836): with pre_select the `elif` never runs, no cell is ever flagged
            flags = np.zeros(len(order), dtype=bool)
            if _batch is not None:
                self._backend.commit(_batch[0], _batch[1], use_invalid=False)
                self._last_invalid_range = np.zeros(_batch[1], dtype=bool)
        elif _batch is not None:
            flags_range = self._backend.mask(_geometries, int(_refine_geometry), first=_batch[0], n=_batch[1])
            self._backend.commit(_batch[0], _batch[1], use_invalid=True)
            flags = flags_range[order - _batch[0]]
            self._last_invalid_range = flags_range
        else:
            flags = self._backend.mask(_geometries, int(_refine_geometry), cells=order)

        # set(filter(None, result)): insertion in iteration order, id 0 and None dropped (s_cube.py:709)
        if self._new_set is set:
            _idx = set(i for i in order[flags].tolist() if i)
        else:
            _idx = IntSet().update_flagged(order, flags)
        if len(_idx) == 0:
            return None
        elif _refine_geometry:
            return _idx
        else:
            invalid = _ordered(_idx)
            self._topo_engine.submit_mark_invalid(invalid)
            if isinstance(self._leaf_cells, IntSet):
                # (== `-= _idx`: discards do not depend on their order; the ids are the new cells just merged in)
                self._leaf_cells.difference_update_ids(invalid)
            else:
                self._leaf_cells -= _idx
            return None

    def refine(self) -> None:
        """generate the grid (s_cube.py:563-667)"""
        logger.info("Starting grid generation.")
        self._refine_uniform()

        iteration_count = 0
        self._n_cells_after_uniform = len(self._leaf_cells)
        if self._n_cells_max is None:
            self._compute_captured_metric()
        self._n_cells_log.append(len(self._leaf_cells))

        logger.info("Starting metric-based refinement.")
        self._times["t_start_adaptive"] = time()
        while self._check_stopping_criteria():
            if self._n_cells_max is None:
                logger.info(f"\r\tStarting iteration no. {iteration_count}, captured metric: "
                            f"{round(self._metric[-1] * 100, 2)} %, N_cells = {len(self._leaf_cells)}")
            else:
                logger.info(f"\r\tStarting iteration no. {iteration_count}, N_cells = {len(self._leaf_cells)}")
            if len(self._metric) >= 2:
                self._compute_n_cells_per_iter()

            # top-N leaves by (gain, -id) -- radix select on the device (replaces heapq.nlargest, s_cube.py:601-602)
            _leaf_cells_sorted = self._backend.topn(self._topo_engine.n_created, min(self._cells_per_iter, self._n_cells))
            to_refine = self._new_set()
            if self._max_delta_level:
                for i in _leaf_cells_sorted.tolist():
                    to_refine.add(i)
                    self._topo.relink_parent_of([i])
                    nb_to_refine_as_well = set(self._check_nb(i))
                    to_refine.update(self._check_constraint(nb_to_refine_as_well))
            else:
                to_refine.update(_leaf_cells_sorted)
                self._topo_engine.submit_relink_parent_of(_leaf_cells_sorted)

            first, n_new = self._refine_cells(to_refine)
            self._remove_invalid_cells(self._new_children, _batch=(first, n_new))

            if self._n_cells_max is None:
                self._compute_captured_metric()
            iteration_count += 1
            self._n_cells_log.append(len(self._leaf_cells))

        if self._n_cells_max is not None:
            self._compute_captured_metric()
        logger.info("Finished metric-based refinement.")

        self._refine_geometries()
        self._update_min_ref_level()
        self._resort_nodes_and_indices_of_grid()
        self._create_mesh_info(iteration_count)
        logger.info(self)
        if self._n_cells_max is not None and self._metric[-1] > 1:
            logger.info("Detected a captured metric > 100%. This means that the current number of 'n_cells_max' can be"
                        " reduced without further loss of information for this metric field, since the metric field is "
                        "over-approximated.")

    # -- 2:1 balance (max_delta_level=True), host only: s_cube.py:447-506 -------------------------------------------
    def _check_nb(self, _cell_no: int) -> list:
        return self._topo.check_nb(_cell_no)

    def _check_constraint(self, nb_violating_constraint: set) -> set:
        new_cells_to_check = True if nb_violating_constraint else False
        while new_cells_to_check:
            tmp = set()
            for c in nb_violating_constraint:
                self._topo.relink_parent_of([c])
                tmp.update(self._check_nb(c))
            if not tmp or tmp.issubset(nb_violating_constraint):
                new_cells_to_check = False
            else:
                nb_violating_constraint.update(tmp)
        return nb_violating_constraint

    # -- geometry refinement: s_cube.py:774-863, 1538-1555 ------------------------------------------------------------
    def _refine_geometries(self) -> None:
        geometries_to_refine = [idx for idx, g in enumerate(self._geometry) if g.refine]
        if geometries_to_refine:
            self._times["t_start_geometry"] = time()
            self._execute_geometry_refinement(_geometries=geometries_to_refine)
            self._times["t_end_geometry"] = time()

    def _execute_geometry_refinement(self, _geometries: list = None) -> None:
        logger.info("Starting geometry refinement.")
        for g in _geometries:
            logger.info(f"Starting refining geometry {self._geometry[g].name}.")
            touching = self._remove_invalid_cells(self._leaf_cells, _refine_geometry=True, _geometry_no=g)
            if touching is None:
                logger.warning("Could not find any cells to refine. Skipping geometry refinement.")
                logger.info("Finished geometry refinement.")
                return
            _all_cells = self._new_set(touching)
            level = self._topo_engine.level_now
            _touching_levels = level[_ordered(_all_cells)]
            _global_min_level = int(_touching_levels.min())
            if self._geometry[g].min_refinement_level is None:
                _global_max_level = int(_touching_levels.max())
            else:
                _global_max_level = self._geometry[g].min_refinement_level
            logger.info(f"Found a minimum cell level of {_global_min_level}. Target level is {_global_max_level}.")

            while _global_max_level > _global_min_level:
                logger.info(f"\r\t\t\t\t\t\t\t\t\tRefining level {_global_min_level+1} / {_global_max_level}.")
                to_refine, checked = self._new_set(), set()
                level = self._topo_engine.level_now
                if self._max_delta_level:
                    for i in _all_cells:
                        if i in checked:
                            continue
                        if int(level[i]) < _global_max_level:
                            to_refine.add(i)
                            self._topo.relink_parent_of([i])
                        nb_to_refine_as_well = set(self._check_nb(i))
                        nb_to_refine_as_well.update(self._check_constraint(nb_to_refine_as_well))
                        to_refine.update(nb_to_refine_as_well)
                        checked.update(nb_to_refine_as_well)
                else:
                    # same insertion sequence as the loop above, neighbour refresh batched into one native call (the
                    # refresh of one parent does not depend on the refresh of another)
                    cells = _ordered(_all_cells)
                    cells = cells[level[cells] < _global_max_level]
                    to_refine.update(cells)
                    self._topo_engine.submit_relink_parent_of(cells)

                first, n_new = self._refine_cells(to_refine)
                _idx_new = self._new_children
                self._remove_invalid_cells(_idx_new, _geometry_no=g, _batch=(first, n_new))

                # among the new *valid* cells, which ones still touch the geometry?  ({i for i in _idx_new if ...}: a
                # new set filled in the iteration order of _idx_new; a new cell is a leaf unless the geometry check above
                # removed it)
                _new_ids = _ordered(_idx_new)
                _new_ids = _new_ids[~self._last_invalid_range[_new_ids - first]]
                still_leaf = self._new_set()
                still_leaf.update(_new_ids.tolist() if self._new_set is set else _new_ids)
                touching = self._remove_invalid_cells(still_leaf, _refine_geometry=True, _geometry_no=g)
                if touching is None:
                    raise TypeError("'NoneType' object is not iterable")    # reference behaviour at s_cube.py:855
                _all_cells = self._new_set(touching)
                _global_min_level += 1

        level = self._topo_engine.level_now
        leaves = _ordered(self._leaf_cells)
        self._current_max_level = int(level[leaves].max())
        logger.info("Finished geometry refinement.")

    # -- final assembly: s_cube.py:734-772 ------------------------------------------------------------------------
    def _resort_nodes_and_indices_of_grid(self) -> None:
        logger.info("Starting renumbering final mesh.")
        self._times["t_start_renumber"] = time()
        dtype = np.int32 if self._n_cells < pt.iinfo(pt.int32).max else np.int64
        faces, nodes = self._topo.finalize(dtype)
        self.face_ids = pt.from_numpy(faces)
        self._final_nodes = pt.from_numpy(nodes)
        centers, levels = self._topo.gather_cells(_ordered(self._leaf_cells))
        self.all_centers = pt.from_numpy(centers)
        self.all_levels = pt.from_numpy(levels).unsqueeze(-1)
        self._times["t_end_renumber"] = time()

    def _create_mesh_info(self, counter: int) -> None:
        """same keys as the reference (s_cube.py:1567-1584)"""
        t = self._times
        self.data_final_mesh["size_initial_cell"] = self._width
        self.data_final_mesh["n_cells_orig"] = self._n_cells_orig
        self.data_final_mesh["n_cells"] = len(self._leaf_cells)
        self.data_final_mesh["iterations"] = counter
        self.data_final_mesh["min_level"] = self._current_min_level
        self.data_final_mesh["max_level"] = self._current_max_level
        self.data_final_mesh["metric_per_iter"] = self._metric
        self.data_final_mesh["cells_per_iter"] = self._n_cells_log
        self.data_final_mesh["t_total"] = t["t_end_renumber"] - t["t_start_uniform"]
        self.data_final_mesh["t_uniform"] = t["t_end_uniform"] - t["t_start_uniform"]
        self.data_final_mesh["t_renumbering"] = t["t_end_renumber"] - t["t_start_renumber"]
        if t["t_end_geometry"] > 0:
            self.data_final_mesh["t_geometry"] = t["t_end_geometry"] - t["t_start_geometry"]
            self.data_final_mesh["t_adaptive"] = t["t_start_geometry"] - t["t_start_adaptive"]
        else:
            self.data_final_mesh["t_geometry"] = None
            self.data_final_mesh["t_adaptive"] = t["t_start_renumber"] - t["t_start_adaptive"]

    def close(self) -> None:
        """release the device arrays, the KNN index and the native topology tables now instead of at garbage collection
        (the result tensors ``all_centers``, ``all_levels``, ``all_nodes``, ``face_ids`` stay valid)"""
        self._backend.close()
        self._topo.close()                        # (waits for a running update first)

    def __len__(self):
        return self._n_cells

    def __str__(self) -> str:
        info = self.data_final_mesh
        message = [f"Finished refinement in {info['t_total']:2.4f} s ", f"({info['iterations']} iterations).",
                   f"Time for uniform refinement: {info['t_uniform']:2.4f} s",
                   f"Time for metric-based refinement: {info['t_adaptive']:2.4f} s"]
        if info['t_geometry'] is not None:
            message += [f"Time for geometry refinement: {info['t_geometry']:2.4f} s"]
        message += ["Time for renumbering the final mesh: {:2.4f} s".format(info['t_renumbering'])]
        message += ["""
                                    Number of cells: {:d}
                                    Minimum ref. level: {:d}
                                    Maximum ref. level: {:d}
                                    Captured metric of original grid: {:.2f} %
                  """.format(len(self._leaf_cells), self._current_min_level, self._current_max_level,
                             self._metric[-1] * 100)]
        return "\n\t\t\t\t\t\t\t\t".join(message)

    @property
    def all_nodes(self):
        """node coordinates: every node created so far while the tree is being refined (the reference keeps a growing
        list, s_cube.py:168,1225), the renumbered ``[n_nodes, d]`` tensor once the grid is assembled (s_cube.py:769)"""
        if self._final_nodes is not None:
            return self._final_nodes
        return pt.from_numpy(self._topo.nodes.copy()) if self._topo is not None else []

    @property
    def n_dimensions(self) -> int:
        return self._n_dimensions

    @property
    def width(self):
        return self._width

    @property
    def geometry(self) -> list:
        return self._geometry

    def _print_settings(self) -> None:
        if self._n_cells_max is not None:
            logger.info("Selecting max. number of cells as stopping criterion.")
        else:
            logger.info("Selecting min. approximation of the metric as stopping criterion.")
        # the reference lists its instance dictionary in creation order (s_cube.py:1669-1692: the stopping criterion in use after the
        # geometry, names padded to the longest ATTRIBUTE name, i.e. one more than the longest shown name); the same lines here, and the
        # backend as an extra last line
        criterion = ("n_cells_max", self._n_cells_max) if self._n_cells_max is not None else ("min_metric", self._min_metric)
        shown = [("pre_select", self._pre_select), ("n_jobs", self._n_jobs), ("max_delta_level", self._max_delta_level),
                 ("geometry", [g.name for g in self._geometry]), criterion, ("min_level", self._min_level),
                 ("cells_per_iter_start", self._cells_per_iter_start), ("cells_per_iter_end", self._cells_per_iter_end),
                 ("cells_per_iter", self._cells_per_iter), ("cells_per_iter_last", self._cells_per_iter_last),
                 ("reach_at_least", self._reach_at_least), ("n_dimensions", self._n_dimensions), ("n_cells_orig", self._n_cells_orig),
                 ("relTol", self._relTol), ("backend", "MI355X / libs3hip.so")]
        width = max(len(k) for k, _ in shown) + 1
        logger.info("\n".join(["\n\tSelected settings:"] + [f"\t\t{k:<{width}}:\t{v}" for k, v in shown]))


def _directions(n_dims: int) -> np.ndarray:
    """offsets of the 2^d children / nodes relative to a cell centre (s_cube.py:188-194)"""
    if n_dims == 2:
        return np.array([[-1, -1], [-1, 1], [1, 1], [1, -1]], dtype=np.float64)
    return np.array([[-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1],
                     [-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1]], dtype=np.float64)


def _initialize_time_dict() -> dict:
    return {"t_start_uniform": 0.0, "t_end_uniform": 0.0, "t_start_adaptive": 0.0,
            "t_start_geometry": 0.0, "t_end_geometry": 0.0, "t_start_renumber": 0.0, "t_end_renumber": 0.0}
