"""dev helper: interleaved A/B, in ONE process, of the order in which the referenced source rows are kept in HBM (ascending point id
vs Hilbert order of the points' coordinates) for the planned interpolation kernel on the bench grid"""
import os, sys, logging, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw); tree.refine()
centers = tree.all_centers.numpy(); tree.close()
k = 26
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3)); idx0, dist = knn.query(centers, k); knn.close()
w = hipops.idw_weights(dist)
out = pt.empty((len(centers), T), dtype=pt.float64, device="cuda")
variants = {}
for name, coords in (("id", None), ("hilbert", x)):
    idx = idx0.clone()
    used, remap = hipops.referenced_rows([idx], len(x), coords=coords)
    hipops.remap_indices(idx, remap)
    plan = hipops.InterpPlan(idx, int(used.numel()), centers); plan.set_weights(w)
    rows = hipops.padded_rows(int(used.numel()), T, pt.float32, "cuda"); rows.normal_()
    variants[name] = (plan, rows)
times = {v: [] for v in variants}
for r in range(9):
    for name, (plan, rows) in variants.items():
        plan.interp(w, rows, out=out); pt.cuda.synchronize()
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.interp(w, rows, out=out)
        e1.record(); pt.cuda.synchronize()
        if r:
            times[name].append(e0.elapsed_time(e1) / 10)
for name, t in times.items():
    print(f"T={T} rows in {name:8s} order: median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}")
