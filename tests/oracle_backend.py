"""
TEST-ONLY compute backend: the interface of ``sparsespatialsampling_amd.tree_backend.HipTreeBackend`` implemented with
the CPU oracle (oracle/s3_oracle.c).  It lets the CPU test-suite exercise the host logic of ``SamplingTree`` (CPython
set ordering, native topology engine, stopping rules) against the reference's golden vectors without a GPU, and it is
the checker the GPU tests compare the HIP backend against.  The product never imports this module.
"""
import numpy as np

from oracle import s3_oracle as orc
from sparsespatialsampling_amd import parallel


class OracleTreeBackend:
    """``grid=True``: the neighbour queries go through the oracle's bucket grid (same results as brute force; the CPU baseline
    of bench.py at the reference's problem sizes)"""
    name = "oracle"

    def __init__(self, vertices, target, k, grid=False):
        self.pts = np.ascontiguousarray(vertices, dtype=np.float64)
        self.y = np.ascontiguousarray(target, dtype=np.float64)
        self.k = int(k)
        self.dim = self.pts.shape[1]
        self.nch = 2 ** self.dim
        self.center = np.zeros((0, self.dim))
        self.level = np.zeros(0, dtype=np.int32)
        self.metric = np.zeros(0)
        self.gain = np.zeros(0)
        self.leaf = np.zeros(0, dtype=bool)
        self.comm = parallel.get_comm()      # N > 1 (gloo on CPU): same split / gather protocol as the HIP backend
        self.grid = orc.GridIndex(self.pts) if grid else None

    def predict(self, q):
        if self.grid is not None:
            return self.grid.idw_predict(self.y, q, self.k)
        return orc.idw_predict(self.pts, self.y, q, self.k)

    def _grow(self, n):
        add = n - len(self.level)
        if add > 0:
            self.center = np.concatenate([self.center, np.zeros((add, self.dim))])
            self.level = np.concatenate([self.level, np.zeros(add, dtype=np.int32)])
            self.metric = np.concatenate([self.metric, np.zeros(add)])
            self.gain = np.concatenate([self.gain, np.zeros(add)])
            self.leaf = np.concatenate([self.leaf, np.zeros(add, dtype=bool)])

    def start(self, root_center, width, gain0, root_metric, root_gain):
        self.width, self.gain0 = width, float(gain0)
        self._grow(1)
        self.center[0] = root_center
        self.metric[0], self.gain[0], self.leaf[0] = root_metric, root_gain, True

    def refine_batch(self, parents, first):
        parents = np.asarray(parents, dtype=np.int64)
        n_new = len(parents) * self.nch
        chunk, b, e = parallel.batch_slice(n_new, self.comm.rank, self.comm.world)
        self._grow(first + chunk * self.comm.world)
        off = (0.25 * self.width) / (2.0 ** self.level[parents])
        ch = self.center[parents][:, None, :] + orc.DIRS[self.dim][None] * off[:, None, None]
        self.center[first:first + n_new] = ch.reshape(n_new, self.dim)
        self.level[first:first + n_new] = np.repeat(self.level[parents] + 1, self.nch)
        if e > b:
            if self.grid is not None:
                m, g = self.grid.child_gain(self.y, self.k, self.center[first + b:first + e],
                                            self.level[first + b:first + e], self.width, self.gain0)
            else:
                m, g = orc.child_gain(self.pts, self.y, self.k, self.center[first + b:first + e],
                                      self.level[first + b:first + e], self.width, self.gain0)
            self.metric[first + b:first + e] = m[:, 0]
            self.gain[first + b:first + e] = g
        if self.comm.world > 1:
            self.comm.allgather_inplace([self.metric[first:], self.gain[first:]], [chunk, chunk])
        self._parents = parents
        self.n_batches = getattr(self, "n_batches", 0) + 1
        return n_new

    def mask(self, geometries, refine_mode, cells=None, first=0, n=None):
        ids = np.asarray(cells, dtype=np.int64) if cells is not None else np.arange(first, first + n)
        c, lv = self.center[ids], self.level[ids]
        inv = np.zeros(len(ids), dtype=bool)
        for g in geometries:
            spec, ki = (g.kernel_spec() if hasattr(g, "kernel_spec") else None), g.keep_inside
            if spec is None:                  # no kernel description: the geometry's own check_cell, cell by cell
                import torch as pt
                from sparsespatialsampling_amd.tree_backend import host_mask
                inv |= host_mask(g, pt.from_numpy(np.ascontiguousarray(c)), pt.from_numpy(np.ascontiguousarray(lv)), self.width,
                                 refine_mode).astype(bool)
            elif spec[0] == "box":
                inv |= orc.mask_box(c, lv, self.width, spec[1], spec[2], refine_mode, ki)
            elif spec[0] == "sphere":
                inv |= orc.mask_sphere(c, lv, self.width, spec[1], spec[2], refine_mode, ki)
            elif spec[0] == "cylinder":
                inv |= _mask_cylinder_spec(c, lv, self.width, spec, refine_mode, ki)
            elif spec[0] == "polygon":
                inv |= orc.mask_polygon(c, lv, self.width, spec[1], refine_mode, ki)
            elif spec[0] == "triangle":
                inv |= orc.mask_triangle(c, lv, self.width, spec[1], refine_mode, ki)
            elif spec[0] == "prism":
                inv |= orc.mask_prism(c, lv, self.width, *spec[1:], refine_mode, ki)
            elif spec[0] == "tetrahedra":
                inv |= orc.mask_tetrahedra(c, lv, self.width, spec[1], spec[2], refine_mode, ki)
            else:
                raise NotImplementedError(spec[0])
        self._last_invalid = inv
        return inv

    def commit(self, first, n_new, use_invalid):
        self.leaf[self._parents] = False
        bad = self._last_invalid if use_invalid else np.zeros(n_new, dtype=bool)
        self.leaf[first:first + n_new] = ~bad
        self.gain[first:first + n_new][bad] = 0.0

    def sumsq(self, n_cells):
        """partial sums of fixed 1024-cell blocks added in block order; metric / leaf are replicated (the all-gather of
        refine_batch), so every rank reduces all blocks itself -- no second collective per iteration (tree_backend.sumsq)"""
        blk = parallel.SUMSQ_BLOCK
        n_blocks = -(-n_cells // blk)
        total = 0.0
        for j in range(n_blocks):
            sl = slice(j * blk, min((j + 1) * blk, n_cells))
            total += orc.sumsq(self.metric[sl][self.leaf[sl]])
        return total

    def topn(self, n_cells, n_top):
        ids = np.flatnonzero(self.leaf[:n_cells])
        return orc.topn(self.gain[ids], ids, n_top)

    def download(self, n_cells):
        return dict(metric=self.metric[:n_cells].copy(), gain=self.gain[:n_cells].copy(),
                    center=self.center[:n_cells].copy(), level=self.level[:n_cells].copy())

    def close(self):
        if self.grid is not None:
            self.grid.close()


def _mask_cylinder_spec(c, lv, width, spec, refine_mode, keep_inside):
    """spec = ("cylinder", p0, axis, norm, r0, r1, is_cone) as produced by CylinderGeometry3D.kernel_spec()"""
    import ctypes as C
    _, p0, axis, norm, r0, r1, cone = spec
    c = np.ascontiguousarray(c, dtype=np.float64)
    lv = np.ascontiguousarray(lv, dtype=np.int32)
    inv = np.empty(len(c), dtype=np.uint8)
    p0 = np.ascontiguousarray(p0, dtype=np.float64)
    axis = np.ascontiguousarray(axis, dtype=np.float64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    orc.lib().s3o_mask_cylinder(P(c), P(lv), C.c_int64(len(c)), C.c_double(float(width)), P(p0), P(axis),
                                C.c_double(float(norm)), C.c_double(float(r0)), C.c_double(float(r1)), int(cone),
                                int(refine_mode), int(keep_inside), P(inv))
    return inv.astype(bool)
