"""dev helper: interleaved A/B in ONE process of the two short-row kernels (weights in registers vs staged in LDS) on a C4-shaped
grid; S3_SHORT_LDS_WEIGHTS is read at every launch"""
import os, sys, logging, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
name = sys.argv[1] if len(sys.argv) > 1 else "box5e7"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
x, metric, geos, kw = bench.build_case(name, dict(bench.WORKLOADS[name]), geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw); tree.refine()
centers = tree.all_centers.numpy(); tree.close()
k = 26
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3)); idx, dist = knn.query(centers, k); knn.close()
w = hipops.idw_weights(dist); del dist
used, remap = hipops.referenced_rows([idx], len(x), coords=x)
hipops.remap_indices(idx, remap); del remap
n = int(used.numel())
plan = hipops.InterpPlan(idx, n, centers); plan.set_weights(w)
rows = hipops.padded_rows(n, T, pt.float32, "cuda"); rows.normal_()
out = pt.empty((len(centers), T), dtype=pt.float64, device="cuda")
times = {"registers": [], "lds": []}
for r in range(9):
    for name_, env in (("registers", None), ("lds", "1")):
        if env: os.environ["S3_SHORT_LDS_WEIGHTS"] = env
        else: os.environ.pop("S3_SHORT_LDS_WEIGHTS", None)
        plan.interp(w, rows, out=out); pt.cuda.synchronize()
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.interp(w, rows, out=out)
        e1.record(); pt.cuda.synchronize()
        if r: times[name_].append(e0.elapsed_time(e1) / 10)
os.environ.pop("S3_SHORT_LDS_WEIGHTS", None)
for name_, t in times.items():
    print(f"{name} T={T} weights in {name_:9s}: median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}")
