"""dev helper: planned (short-row / chunk-pipelined) vs direct interpolation kernel over snapshot-batch lengths on the
cylinder3D bench grid.  S3_SHORT_ROW_CHUNKS (read once per process) moves the switch between the two planned variants.
    python tools/rowlen_probe.py [T ...]"""
import sys, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
ts = [int(a) for a in sys.argv[1:]] or [16, 25, 32, 48, 64, 100, 128, 200, 256, 400, 1000]
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
k = 26
knn = hipops.KnnIndex(x); idx, dist = knn.query(centers, k); knn.close()
w = hipops.idw_weights(dist)
nc, n = len(centers), len(x)
nu = int(pt.unique(idx).numel())
plan = hipops.InterpPlan(idx, n, centers)
print(f"cells {nc} points {n} unique rows {nu} tiles {plan.n_tiles} staged rows {plan.total_rows}", flush=True)


def timed(fn, reps=10):
    fn(); pt.cuda.synchronize()
    e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); pt.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for t in ts:
    data = hipops.padded_rows(n, t, pt.float32, "cuda"); data.normal_()
    dense = data.contiguous()
    out = pt.empty((nc, t), dtype=pt.float64, device="cuda")
    ms_p = timed(lambda: plan.interp(w, data, out=out))
    got = out.clone()
    ms_d = timed(lambda: hipops.interp(w, idx, dense, out=out))
    same = bool(pt.equal(got, out))
    balg = nu * t * 4 + nc * t * 8 + nc * k * 12
    print(f"T={t:5d} pitch {data.stride(0) * 4:5d} B  planned {ms_p:7.3f} ms = {nc * t / ms_p / 1e6:6.1f} G/s, {balg / ms_p / 1e6:5.0f} GB/s alg "
          f"({balg / ms_p / 8e9:.3f} of peak) | direct {ms_d:7.3f} ms ({balg / ms_d / 8e9:.3f}) | bit-equal {same}", flush=True)
    del data, dense, out, got
