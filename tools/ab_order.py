"""dev helper: interleaved A/B timing, in ONE process, of launch-time switches of the planned interpolation on the bench's
headline shape -- a dense [N, T] fp32 batch read in place (s3_interp_planned_src) on the cylinder3D grid.  Every variant's
result is compared bit for bit with the first one's.
    python tools/ab_order.py "base:;split4:S3_PLAN_SPLIT=4;brick64x4:S3_PLAN_SPLIT=4,S3_PLAN_BRICK=64"  [T] [rounds] [launches per round]
AB_PITCHED=1: the pitched, compacted copy of the referenced rows instead (s3_interp_planned)."""
import os, sys, logging, statistics
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
spec = sys.argv[1] if len(sys.argv) > 1 else "base:"
t = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
name = os.environ.get("AB_WORKLOAD", "cylinder3D_Re3900")
cfg = dict(bench.WORKLOADS[name])
x, metric, geos, kw = bench.build_case(name, cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
k, n = 26, len(x)
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
idx, dist = knn.query(centers, k)
w = hipops.idw_weights(dist)
knn.close()
used, remap = hipops.referenced_rows([idx], n, coords=x)
hipops.remap_indices(idx, remap)
plan = hipops.InterpPlan(idx, int(used.numel()), centers)
plan.set_weights(w)
plan.set_source_ids(used.contiguous(), n)
nc = len(centers)
print(f"{n} points, {nc} cells, {int(used.numel())} referenced rows, {plan.n_tiles} tiles, T={t}", flush=True)
table = pt.empty((n, t), dtype=pt.float32, device="cuda").normal_(generator=pt.Generator(device="cuda").manual_seed(t))
out = pt.empty((nc, t), dtype=pt.float64, device="cuda")
if os.environ.get("AB_PITCHED") == "1":
    rows = hipops.gather_rows(table, used.contiguous(), hipops.padded_rows(int(used.numel()), t, pt.float32, "cuda"))
    del table
    launch = lambda: plan.interp(w, rows, out=out)
else:
    launch = lambda: plan.interp_src(table, out=out)
variants = []
for item in spec.split(";"):
    vname, _, envs = item.partition(":")
    variants.append((vname, dict(e.split("=") for e in envs.split(",") if e)))
ref = None
for vname, env in variants:
    os.environ.update(env)
    out.zero_()
    launch()
    pt.cuda.synchronize()
    for kk in env:
        del os.environ[kk]
    if ref is None:
        ref = out.clone()
    else:
        print(f"{vname}: same bits as {variants[0][0]}: {bool(pt.equal(out, ref))}", flush=True)
del ref
times = {v[0]: [] for v in variants}
for r in range(rounds + 1):
    for vname, env in variants:
        os.environ.update(env)
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        launch()
        e0.record()
        for _ in range(reps):
            launch()
        e1.record(); pt.cuda.synchronize()
        for kk in env:
            del os.environ[kk]
        if r:
            times[vname].append(e0.elapsed_time(e1) / reps)
b_alg = int(used.numel()) * t * 4 + nc * t * 8 + nc * k * 12
for vname, tt in times.items():
    med = statistics.median(tt)
    print(f"T={t}: {vname:28s} median {med:.4f} ms  min {min(tt):.4f}  max {max(tt):.4f}   frac of 8 TB/s {b_alg / (med * 1e-3) / 8e12:.3f}", flush=True)
