"""dev helper: one leaf-cell shard of 8 (cylinder3D grid): launch time of the planned kernel against the smallest grid size at
which the column chunks are split over blockIdx.y (S3_PLAN_MIN_BLOCKS, read once per process -> one subprocess per value)"""
import os, subprocess, sys
if len(sys.argv) == 1:
    for mb in (1, 2048, 4096, 8192, 16384, 40000):
        env = dict(os.environ, S3_PLAN_MIN_BLOCKS=str(mb))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"S3_PLAN_MIN_BLOCKS={mb}: " + (r.stdout.strip().splitlines() or [r.stderr[-500:]])[-1], flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, hipops, parallel
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw); tree.refine()
centers = tree.all_centers.numpy(); tree.close()
k = 26
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
res = []
for world, rank in ((8, 1), (8, 3), (1, 0)):
    mine = centers
    if world > 1:
        mine = np.ascontiguousarray(centers[parallel.LeafShards(knn, centers, k, rank, world).mine])
    idx, dist = knn.query(mine, k)
    w = hipops.idw_weights(dist)
    used, remap = hipops.referenced_rows([idx], len(x), coords=x)
    hipops.remap_indices(idx, remap)
    n = int(used.numel())
    plan = hipops.InterpPlan(idx, n, mine); plan.set_weights(w)
    for T in (1000, 200):
        data = hipops.padded_rows(n, T, pt.float32, "cuda"); data.normal_()
        out = pt.empty((len(mine), T), dtype=pt.float64, device="cuda")
        for _ in range(3): plan.interp(w, data, out=out)
        pt.cuda.synchronize()
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.interp(w, data, out=out)
        e1.record(); pt.cuda.synchronize()
        res.append(f"W{world}r{rank} T{T} ({plan.n_tiles} tiles) {e0.elapsed_time(e1) / 20:.3f} ms")
        del data, out
    del plan
print("; ".join(res))
