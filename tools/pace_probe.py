"""dev probe: one launch at a time of the in-place headline kernel under S3_PACE_TICKS values, each timed and printed at once
    python tools/pace_probe.py 0 450 520 600"""
import os, sys, logging, time
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
name = os.environ.get("AB_WORKLOAD", "cylinder3D_Re3900")
cfg = dict(bench.WORKLOADS[name])
x, metric, geos, kw = bench.build_case(name, cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
k, n, t = 26, len(x), int(os.environ.get("AB_T", "1000"))
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
idx, dist = knn.query(centers, k)
w = hipops.idw_weights(dist)
knn.close()
used, remap = hipops.referenced_rows([idx], n, coords=x)
hipops.remap_indices(idx, remap)
plan = hipops.InterpPlan(idx, int(used.numel()), centers)
plan.set_weights(w)
plan.set_source_ids(used.contiguous(), n)
table = pt.empty((n, t), dtype=pt.float32, device="cuda").normal_()
out = pt.empty((len(centers), t), dtype=pt.float64, device="cuda")
plan.interp_src(table, out=out)
pt.cuda.synchronize()
ref = out.clone()
print(f"{plan.n_tiles} tiles", flush=True)
for p in sys.argv[1:]:
    os.environ["S3_PACE_TICKS"] = p
    for rep in range(3):
        out.zero_()
        pt.cuda.synchronize()
        t0 = time.perf_counter()
        plan.interp_src(table, out=out)
        pt.cuda.synchronize()
        print(f"pace {p}: launch {rep}: {(time.perf_counter() - t0) * 1e3:.3f} ms  same bits {bool(pt.equal(out, ref))}", flush=True)
