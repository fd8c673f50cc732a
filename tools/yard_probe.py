"""dev probe: the headline launch next to its yardsticks, interleaved in one process -- hand-written float4 copy,
7-read / 2-write mix, reads only (s3_yard_stream), and the LOADS of the headline's tile plan on the headline's table with one line /
two consecutive lines of a row per visit (s3_yard_plan_loads).  Under `rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum ...` the
kernel names tell the variants apart (plan_loads_kernel<0> / <1>, interp_planned_shift_kernel).
    python tools/yard_probe.py [rounds]"""
import os, sys, logging, statistics
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only_plan = len(sys.argv) > 2 and sys.argv[2] == "plan"          # (under a PMC pass: the plan legs and the headline only)
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
plan = hipops.InterpPlan(idx, int(used.numel()), centers)
plan.set_weights(w)
plan.set_source_ids(used.contiguous(), n)
data = pt.empty((n, t), dtype=pt.float32, device="cuda").normal_()
out = pt.empty((len(centers), t), dtype=pt.float64, device="cuda")
rows_p = hipops.gather_rows(data, used.contiguous(), hipops.padded_rows(int(used.numel()), t, pt.float32, "cuda"))
src = pt.empty((4 << 30) // 4, dtype=pt.float32, device="cuda").normal_()
dst = pt.empty((4 << 30) // 4, dtype=pt.float32, device="cuda")
b_alg = int(used.numel()) * t * 4 + len(centers) * t * 8 + len(centers) * k * 12
legs = {
    "headline (in place)": (lambda: plan.interp_src(data, out=out), b_alg),
    "headline (pitched copy)": (lambda: plan.interp(w, rows_p, out=out), b_alg),
    "plan loads, 1 line / visit, in place": (lambda: plan.yard_loads(data, 0), plan.yard_loads(data, 0)),
    "plan loads, 2 lines / visit, in place": (lambda: plan.yard_loads(data, 1), plan.yard_loads(data, 1)),
    "plan loads, 1 line / visit, pitched": (lambda: plan.yard_loads(rows_p, 0, in_place=False), plan.yard_loads(rows_p, 0, in_place=False)),
    "plan loads, 2 lines / visit, pitched": (lambda: plan.yard_loads(rows_p, 1, in_place=False), plan.yard_loads(rows_p, 1, in_place=False)),
}
for key, r_, w_ in (() if only_plan else (("copy float4", 1, 1), ("mix 7r2w", 7, 2), ("read only", 4, 0))):
    for nt in (False, True):
        legs[f"{key}{' nt' if nt else ''}"] = ((lambda r_=r_, w_=w_, nt=nt: hipops.yard_stream(src, dst, r_, w_, nontemporal=nt)),
                                               sum(hipops.yard_stream(src, dst, r_, w_, nontemporal=nt)))
if not only_plan:
    legs["torch dst.copy_(src) (runtime blit)"] = (lambda: dst.copy_(src), 2 * src.numel() * 4)
times = {k_: [] for k_ in legs}
for r in range(rounds + 1):
    for key, (fn, _) in legs.items():
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(4):
            fn()
        e1.record(); pt.cuda.synchronize()
        if r:
            times[key].append(e0.elapsed_time(e1) / 4)
for key, (_, nbytes) in legs.items():
    med = statistics.median(times[key])
    print(f"{key:45s} median {med:8.4f} ms  min {min(times[key]):8.4f}  {nbytes / 1e9:7.2f} GB  {nbytes / med / 1e9:6.3f} TB/s", flush=True)
