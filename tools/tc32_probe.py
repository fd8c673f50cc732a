"""dev probe (r6): 32-cell tiles (four 128-thread workgroups per CU, 240-row LDS images) against the default 64-cell tiles on the
pitched, compacted copy of the referenced rows -- the chunk kernel `interp_planned_kernel<T, TC>`; bit-equality first, then
interleaved timing in one process.  Needs a build that accepts tile_cells = 32 (a three-line patch of csrc/interp_plan.hip, measured and
NOT kept: HISTORY 10, profiles/r06/extra/tc32_probe.txt).     python tools/tc32_probe.py [rounds]"""
import os, sys, logging, statistics
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
k, n, t = 26, len(x), 1000
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
idx, dist = knn.query(centers, k)
w = hipops.idw_weights(dist)
knn.close()
used, remap = hipops.referenced_rows([idx], n, coords=x)
hipops.remap_indices(idx, remap)
n_rows = int(used.numel())
rows_p = hipops.padded_rows(n_rows, t, pt.float32, "cuda")
rows_p.normal_(generator=pt.Generator(device="cuda").manual_seed(3))
plans = {tc: hipops.InterpPlan(idx, n_rows, centers, tile_cells=tc) for tc in (64, 32)}
outs = {tc: pt.empty((len(centers), t), dtype=pt.float64, device="cuda") for tc in plans}
for tc, p in plans.items():
    p.set_weights(w)
    p.interp(w, rows_p, out=outs[tc])
    print(f"tc {tc}: tiles {p.n_tiles}, staged rows {p.total_rows}", flush=True)
pt.cuda.synchronize()
print("same bits:", bool(pt.equal(outs[64], outs[32])), flush=True)
b_alg = n_rows * t * 4 + len(centers) * t * 8 + len(centers) * k * 12
times = {tc: [] for tc in plans}
for r in range(rounds + 1):
    for tc, p in plans.items():
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        p.interp(w, rows_p, out=outs[tc])
        e0.record()
        for _ in range(8):
            p.interp(w, rows_p, out=outs[tc])
        e1.record(); pt.cuda.synchronize()
        if r:
            times[tc].append(e0.elapsed_time(e1) / 8)
for tc, v in times.items():
    med = statistics.median(v)
    print(f"tc {tc}: median {med:.4f} ms  min {min(v):.4f}  frac of 8 TB/s {b_alg / (med * 1e-3) / 8e12:.3f}", flush=True)
