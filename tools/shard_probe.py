"""dev helper: launch time of the planned kernel for one leaf-cell shard of W (what one rank of an N-GPU run executes per step),
on one GPU: predicts the strong-scaling curve of `bench.py --gpus N` up to launch / synchronisation overheads"""
import os, sys, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, hipops, parallel
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
T = 1000
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw); tree.refine()
centers = tree.all_centers.numpy(); tree.close()
k, nc_total = 26, len(centers)
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
# S3_PROBE_T_SPLIT=2: at 8 ranks the hybrid decomposition of bench.py (4 leaf-cell shards x 2 halves of the snapshot axis)
t_split = int(os.environ.get("S3_PROBE_T_SPLIT", "1"))
for world_total in [int(v) for v in os.environ.get("S3_PROBE_WORLDS", "1,2,4,8").split(",")]:
    wt = t_split if world_total >= 8 else 1
    world, T = world_total // wt, 1000 // wt
    worst = 0.0
    for rank in range(world):
        if world > 1 and os.environ.get("S3_PROBE_CREATION_ORDER") != "1":
            sh = parallel.LeafShards(knn, centers, k, rank, world)      # cost-balanced compact shards (what bench.py / ExportData use)
            mine = np.ascontiguousarray(centers[sh.mine])
        else:
            c0, c1 = parallel.shard_range(nc_total, rank, world)        # equal counts in creation order (round-2 start)
            mine = np.ascontiguousarray(centers[c0:c1])
        idx, dist = knn.query(mine, k)
        w = hipops.idw_weights(dist)
        used, remap = hipops.referenced_rows([idx], len(x), coords=x)
        hipops.remap_indices(idx, remap)
        n = int(used.numel())
        plan = hipops.InterpPlan(idx, n, mine); plan.set_weights(w)
        data = hipops.padded_rows(n, T, pt.float32, "cuda"); data.normal_()
        out = pt.empty((len(mine), T), dtype=pt.float64, device="cuda")
        for _ in range(3): plan.interp(w, data, out=out)
        pt.cuda.synchronize()
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.interp(w, data, out=out)
        e1.record(); pt.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        worst = max(worst, ms)
        print(f"  W={world} rank {rank}: {len(mine)} cells, {n} resident rows, {plan.total_rows} staged rows, {plan.n_tiles} tiles: {ms:.3f} ms", flush=True)
        del plan, data, out, idx, w
    print(f"W={world_total} ({world} cell shards x {wt} snapshot parts): slowest shard {worst:.3f} ms -> {nc_total * 1000 / worst / 1e6:.0f} G cell*snapshots/s whole job")
knn.close()
