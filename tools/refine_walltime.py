"""dev helper: wall-clock per hipops / IntSet / topology call inside one SamplingTree.refine() of a bench workload (no
profiler: plain perf_counter wrappers), to see where the host blocks"""
import os, sys, time, logging, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, hipops, s_cube, tree_backend, intset
logging.getLogger().setLevel(logging.WARNING)
acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[label]
            e[0] += 1
            e[1] += time.perf_counter() - t0
    setattr(obj, name, timed)


for n in ("make_children", "child_gain", "mask_box", "mask_cylinder", "commit_batch", "sumsq_leaf", "topn_leaf", "topn_scratch", "to_device"):
    wrap(hipops, n, "hipops." + n)
for n in ("update_flagged", "__isub__", "to_array", "difference_update_ids", "__len__"):
    wrap(intset.IntSet, n, "IntSet." + n)
wrap(intset.RangeSet, "to_array", "RangeSet.to_array")
_update = intset.IntSet.update


def update_by_kind(self, items):
    t0 = time.perf_counter()
    try:
        return _update(self, items)
    finally:
        e = acc["IntSet.update(" + type(items).__name__ + (", deferred)" if self._deferred else ")")]
        e[0] += 1
        e[1] += time.perf_counter() - t0


intset.IntSet.update = update_by_kind
for n in ("submit_refine", "submit_relink_parent_of", "submit_mark_invalid", "sync", "finalize", "gather_cells"):
    wrap(s_cube._Topology, n, "topo." + n)
for n in ("finalize", "gather_cells", "sync"):
    wrap(s_cube._DeviceTopology, n, "devtopo." + n)
wrap(hipops, "to_host", "hipops.to_host")
for n in ("mask", "refine_batch", "topn", "sumsq", "commit"):
    wrap(tree_backend.HipTreeBackend, n, "backend." + n)
for n in ("_remove_invalid_cells", "_refine_cells", "_compute_captured_metric", "_refine_uniform", "_refine_geometries", "_resort_nodes_and_indices_of_grid"):
    wrap(s_cube.SamplingTree, n, "tree." + n)

name = sys.argv[1] if len(sys.argv) > 1 else "cylinder3D_Re3900"
x, metric, geos, tree_kw = bench.build_case(name, dict(bench.WORKLOADS[name]), geometry)
for rep in range(2):
    acc.clear()
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **tree_kw)
    pt.cuda.synchronize()
    t0 = time.perf_counter()
    tree.refine()
    pt.cuda.synchronize()
    print(f"rep {rep}: refine {time.perf_counter() - t0:.4f} s", {k: (None if v is None else round(v, 4)) for k, v in tree.data_final_mesh.items() if k.startswith("t_")})
    if hasattr(tree._topo_engine, "_lib") and hasattr(tree._topo_engine, "_h"):           # the host topology engine only
        import ctypes as C
        st = np.zeros(12)
        tree._topo_engine._lib.s3t_stats(tree._topo_engine._h, st.ctypes.data_as(C.c_void_p))
        names = ["validate", "passA", "scan", "passC", "passD", "finish", "relink pass", "relink_parent_of", "mark_invalid", "growth", "sequential batches"]
        print("  engine phases [ms]:", {n: round(v * 1e3, 2) for n, v in zip(names, st)}, "sum", round(st.sum() * 1e3, 1))
    tree.close()
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:42s} {n:5d} calls {t * 1e3:9.2f} ms")
