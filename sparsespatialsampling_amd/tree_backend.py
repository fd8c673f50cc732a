"""
Device side of the S^3 sampling tree: the structure-of-arrays image of the reference's ``Cell`` objects
(s_cube.py:32-83) lives in HBM and every numerical step of ``SamplingTree.refine`` is one libs3hip.so call.

HBM-resident arrays (capacity grows geometrically; sized for 288 GB, one cell costs 8*dim + 4 + 8 + 8 + 1 bytes):
    center [cap, dim] f64, level [cap] i32, metric [cap] f64, gain [cap] f64, leaf [cap] u8
plus the bucket-grid KNN index over the original points and the metric in bucket order (hipops.KnnIndex).

Per refine batch the host sends the ordered parent ids (4 B each) and receives the invalid flags of the new cells
(1 B each); per iteration it receives the ordered top-N ids and one double (sum of metric^2 over the leaves).
"""
import numpy as np
import torch as pt

from . import hipops, parallel


def _level_factor_table(width, n_dims, n_levels=64):
    """1/2^d * (width/2^level)^d, the reference's Python expression at s_cube.py:1859, evaluated on the host so the
    device multiplies by bit-identical factors."""
    return np.array([1 / (2 ** n_dims) * ((width / (2 ** lv)) ** n_dims) for lv in range(n_levels)], dtype=np.float64)


def host_mask(geometry, center, level, width, refine_mode):
    """verdicts of ``geometry.check_cell`` for the cells with the given centres ``[n, d]`` / levels ``[n]`` (CPU tensors)
    -> uint8 [n]: the path of a geometry without a device predicate.  The node coordinates are the reference's
    ``centre + direction * 0.5 * width / 2^level`` (s_cube.py:441, factor 0.5; all scalings exact), ``[2^d, d]`` per cell."""
    from .s_cube import _directions
    n, dim = int(center.shape[0]), int(center.shape[1])
    dirs = pt.from_numpy(np.asarray(_directions(dim), dtype=np.float64))                     # [2^d, d]
    half = (0.5 * float(width)) / (2.0 ** level.double())                                     # [n]
    nodes = center.double()[:, None, :] + dirs[None, :, :] * half[:, None, None]             # [n, 2^d, d]
    return np.fromiter((bool(geometry.check_cell(nodes[i], bool(refine_mode))) for i in range(n)), dtype=np.uint8, count=n)


class HipTreeBackend:
    name = "hip"

    def __init__(self, vertices, target, k):
        self.dev = hipops.device()
        self.k = int(k)
        self.knn = hipops.KnnIndex(vertices, hipops.knn_occupancy(int(k), int(vertices.shape[1])))
        self.knn.set_values(target)
        self.dim = self.knn.dim
        self.nch = 2 ** self.dim
        self.cap = 0
        self.center = self.level = self.metric = self.gain = self.leaf = self.child_metric = None
        self._parents = None
        self._poly_cache = {}
        self.comm = parallel.get_comm()          # more than one rank: the KNN work of a batch is split, see refine_batch

    # -- plain KNN regression at arbitrary points (root cell, s_cube.py:372) ------------------------------------
    def predict(self, q):
        return self.knn.predict(np.ascontiguousarray(q, dtype=np.float64), self.k).cpu().numpy()

    # -- cell arrays -------------------------------------------------------------------------------------------
    def _grow(self, need):
        if need <= self.cap:
            return
        cap = max(4096, self.cap)
        while cap < need:
            cap *= 2
        new = dict(center=pt.empty((cap, self.dim), dtype=pt.float64, device=self.dev),
                   level=pt.zeros(cap, dtype=pt.int32, device=self.dev),
                   metric=pt.zeros(cap, dtype=pt.float64, device=self.dev),
                   gain=pt.zeros(cap, dtype=pt.float64, device=self.dev),
                   leaf=pt.zeros(cap, dtype=pt.uint8, device=self.dev),
                   # predictions at every cell's 2^d candidate child centres: a child's centre value when it is created
                   child_metric=pt.empty((cap, self.nch), dtype=pt.float64, device=self.dev))
        if self.cap:
            for name, t in new.items():
                t[:self.cap].copy_(getattr(self, name)[:self.cap])
        for name, t in new.items():
            setattr(self, name, t)
        self.cap = cap
        n_blocks = -(-cap // parallel.SUMSQ_BLOCK)
        self._sumsq_partial = pt.zeros(n_blocks, dtype=pt.float64, device=self.dev)
        self._sumsq_out = pt.empty(1, dtype=pt.float64, device=self.dev)

    def start(self, root_center, width, gain0, root_metric, root_gain):
        self.width = width
        self.gain0 = float(gain0)
        self.level_factor = hipops.to_device(_level_factor_table(width, self.dim))
        self._grow(4096)
        self.center[0].copy_(pt.from_numpy(np.asarray(root_center, dtype=np.float64)))
        self.level[0] = 0
        self.metric[0] = float(root_metric)
        self.gain[0] = float(root_gain)
        self.leaf[0] = 1
        # the root's own row of child_metric (the centre values of its children), so that the first batch already takes the
        # 8-query wavefront route: the root as a batch of one cell, its metric and gain (known from the host) written aside
        # (the call looks the root's "centre value" up in child_metric[parent = 0][0] -- a row it is about to write: zeroed first,
        # so that nothing uninitialised is read; the metric and gain it derives from it go aside and are dropped)
        self.child_metric[0].zero_()
        aside = pt.zeros(2, dtype=pt.float64, device=self.dev)
        scratch = pt.empty((self.nch + 1) + 2 + self.nch, dtype=pt.float64, device=self.dev)
        hipops.child_gain_reuse(self.knn, self.k, self.center, self.level, 0, 1, float(self.width), self.level_factor,
                                self.gain0, aside[:1], aside[1:], scratch, pt.zeros(1, dtype=pt.int32, device=self.dev), 0,
                                self.child_metric)

    def refine_batch(self, parents, first):
        """children of the ordered ``parents`` become cells first..first+len*2^d-1; their metric and gain are
        evaluated (s_cube.py:875-900 -> a3 + a4)."""
        n_par = len(parents)
        n_new = n_par * self.nch
        chunk, b, e = parallel.batch_slice(n_new, self.comm.rank, self.comm.world)
        self._grow(first + chunk * self.comm.world)              # room for the gathered slices (equal-sized chunks)
        self._parents = hipops.to_device(np.ascontiguousarray(parents, dtype=np.int32))
        hipops.make_children(self.center, self.level, self._parents, first, float(self.width))
        # the KNN metric / gain of this rank's slice of the new cells (all of them with one rank) ...
        if e > b:
            scratch = pt.empty((e - b) * (self.nch + 1) + 2 + (e - b) * self.nch, dtype=pt.float64, device=self.dev)
            # a new cell's centre is a point its parent's call predicted already: only the 2^d child points are searched
            # (8 of 9 queries in 3-D), by the wavefront kernels (the root's row was filled by start()).  With several ranks
            # the rows of child_metric travel with the metric and the gain below, so a parent's entry is at hand whichever
            # rank computed it.
            hipops.child_gain_reuse(self.knn, self.k, self.center, self.level, first + b, e - b, float(self.width),
                                    self.level_factor, self.gain0, self.metric, self.gain, scratch, self._parents, b,
                                    self.child_metric)
        # ... and one grouped all-gather hands every rank the others' slices
        if self.comm.world > 1:
            self.comm.allgather_inplace([self.metric[first:], self.gain[first:], self.child_metric[first:].view(-1)],
                                        [chunk, chunk, chunk * self.nch])
        return n_new

    def mask(self, geometries, refine_mode, cells=None, first=0, n=None):
        """OR of the geometry verdicts (s_cube.py:1831-1837) for the cells ``cells`` (ordered id array) or the id
        range first..first+n-1 -> numpy bool [n]."""
        d_cells = None
        if cells is not None:
            n = len(cells)
            d_cells = hipops.to_device(np.ascontiguousarray(cells, dtype=np.int32))
        invalid = pt.zeros(max(n, 1), dtype=pt.uint8, device=self.dev)
        w = float(self.width)
        for g in geometries:
            spec = g.kernel_spec() if hasattr(g, "kernel_spec") else None
            ki = int(g.keep_inside)
            if spec is None:
                # no device predicate (a user-defined geometry): the reference's own protocol, check_cell(nodes, refine_geometry)
                # per cell (s_cube.py:1816-1837), on the downloaded node coordinates
                flags = self._mask_on_host(g, bool(refine_mode), d_cells, first, n)
                invalid[:n] |= hipops.to_device(flags)
            elif spec[0] == "box":
                hipops.mask_box(self.center, self.level, d_cells, first, n, w, spec[1], spec[2], refine_mode, ki, invalid)
            elif spec[0] == "sphere":
                hipops.mask_sphere(self.center, self.level, d_cells, first, n, w, spec[1], spec[2], refine_mode, ki,
                                   invalid)
            elif spec[0] == "cylinder":
                hipops.mask_cylinder(self.center, self.level, d_cells, first, n, w, *spec[1:], refine_mode, ki, invalid)
            elif spec[0] == "polygon":
                key = id(g)
                if key not in self._poly_cache:
                    self._poly_cache[key] = hipops.to_device(np.ascontiguousarray(spec[1], dtype=np.float64))
                hipops.mask_polygon(self.center, self.level, d_cells, first, n, w, self._poly_cache[key], refine_mode,
                                    ki, invalid)
            elif spec[0] == "triangle":
                hipops.mask_triangle(self.center, self.level, d_cells, first, n, w, spec[1], refine_mode, ki, invalid)
            elif spec[0] == "prism":
                hipops.mask_prism(self.center, self.level, d_cells, first, n, w, *spec[1:], refine_mode, ki, invalid)
            elif spec[0] == "tetrahedra":
                hipops.mask_tetrahedra(self.center, self.level, d_cells, first, n, w, spec[1], spec[2], refine_mode, ki,
                                       invalid)
            else:
                raise NotImplementedError(f"geometry kind {spec[0]!r} has no device kernel")
        self._last_invalid = invalid
        return invalid[:n].cpu().numpy().astype(bool)

    def _mask_on_host(self, geometry, refine_mode, d_cells, first, n):
        if d_cells is not None:
            center, level = self.center[d_cells.long()].cpu(), self.level[d_cells.long()].cpu()
        else:
            center, level = self.center[first:first + n].cpu(), self.level[first:first + n].cpu()
        return host_mask(geometry, center, level, self.width, refine_mode)

    def commit(self, first, n_new, use_invalid):
        """leaf/gain bookkeeping of the batch created by the last ``refine_batch`` (s_cube.py:250-251, 721-723)."""
        hipops.commit_batch(self.leaf, self.gain, self._parents, first, n_new,
                            self._last_invalid if use_invalid else None)

    def sumsq(self, n_cells):
        """sum of metric^2 over the leaves: partial sums of fixed 1024-cell blocks added in block order (a fixed tree, so the
        bits do not depend on who computes them).  ``metric`` and ``leaf`` are replicated -- the grouped all-gather of
        ``refine_batch`` has handed every rank all slices, ``commit`` runs on every rank -- so every rank reduces ALL blocks
        itself (9 MB at cylinder3D's size, 100 MB at 5*10^7 points: microseconds) and no second collective per iteration is
        needed: the all-gather of the batch is the one exchange of a refinement step (reference s_cube.py:317-336)"""
        n_blocks = -(-n_cells // parallel.SUMSQ_BLOCK)
        hipops.sumsq_blocks(self.metric, self.leaf, n_cells, 0, n_blocks, self._sumsq_partial)
        hipops.sum_ordered(self._sumsq_partial, n_blocks, self._sumsq_out)
        return float(self._sumsq_out.item())

    def topn(self, n_cells, n_top):
        scratch = hipops.topn_scratch(n_cells, n_top, self.dev)
        return hipops.topn_leaf(self.gain, self.leaf, n_cells, n_top, scratch).astype(np.int64)

    def download(self, n_cells):
        return dict(metric=self.metric[:n_cells].cpu().numpy(), gain=self.gain[:n_cells].cpu().numpy(),
                    center=self.center[:n_cells].cpu().numpy(), level=self.level[:n_cells].cpu().numpy())

    def close(self):
        """release the KNN index and every device array now (the owning tree may sit in a reference cycle that the
        garbage collector only breaks later)"""
        self.knn.close()
        self.center = self.level = self.metric = self.gain = self.leaf = self.child_metric = None
        self._parents = self._last_invalid = None
        self._sumsq_out = self._sumsq_partial = self.level_factor = None
        self._poly_cache = {}
