"""dev helper: interpolation kernel time for small snapshot batches (direct gather vs planned), bench grid"""
import sys, logging, statistics
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
tree.refine()
centers = tree.all_centers.numpy()
k = 26
idx, dist = hipops.KnnIndex(x).query(centers, k)
w = hipops.idw_weights(dist)
nc, n = len(centers), len(x)
plan = hipops.InterpPlan(idx, n, centers)
def timeit(f, reps=10):
    f(); pt.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record(); pt.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
    return statistics.median(ts)
for T in (16, 24, 25, 32, 50, 64, 100, 128, 256):
    dense = pt.randn((n, 1, T), dtype=pt.float32, device="cuda")
    out = pt.empty((nc, 1, T), dtype=pt.float64, device="cuda")
    t_direct = timeit(lambda: hipops.interp(w, idx, dense, out=out))
    line = f"T={T:4d}: direct {t_direct:7.3f} ms = {nc*T/t_direct/1e6:7.1f} G/s"
    rows = hipops.padded_rows(n, T, pt.float32, "cuda"); rows.copy_(dense.reshape(n, T))
    if plan.supports(k, rows):
        out2 = pt.empty((nc, T), dtype=pt.float64, device="cuda")
        t_plan = timeit(lambda: plan.interp(w, rows, out=out2))
        assert pt.equal(out2, out.reshape(nc, T))
        line += f"   planned {t_plan:7.3f} ms = {nc*T/t_plan/1e6:7.1f} G/s"
    print(line, flush=True)
    del dense, out, rows
