"""dev helper: the interpolation of a device-resident dense batch [N, T] fp32 READ IN PLACE (s3_interp_planned_src) on the
cylinder3D bench grid -- interleaved in one process: whole aligned lines with the per-row phase undone on the way into LDS
(default) against straddling segments (S3_INPLACE_SHIFT=0) against the planned kernel on the pitched, compacted copy of the
referenced rows (the layout ExportData uploads host batches into).  Checks the three against the direct gather kernel.
    python tools/ab_inplace.py [T ...]      (AB_ROUNDS, AB_REPS)"""
import os, sys, logging, statistics
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
ts = [int(a) for a in sys.argv[1:]] or [1000]
rounds, reps = int(os.environ.get("AB_ROUNDS", "7")), int(os.environ.get("AB_REPS", "10"))
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
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
idx_c = idx.clone()
hipops.remap_indices(idx_c, remap)
plan = hipops.InterpPlan(idx_c, int(used.numel()), centers)
plan.set_weights(w)
plan.set_source_ids(used.contiguous(), n)
nc = len(centers)
print(f"{n} points, {nc} cells, {int(used.numel())} referenced rows, {plan.n_tiles} tiles", flush=True)
for t in ts:
    table = pt.empty((n, t), dtype=pt.float32, device="cuda").normal_(generator=pt.Generator(device="cuda").manual_seed(t))
    rows = hipops.gather_rows(table, used.contiguous(), hipops.padded_rows(int(used.numel()), t, pt.float32, "cuda"))
    out = pt.empty((nc, t), dtype=pt.float64, device="cuda")
    variants = [("in place, aligned lines", {"S3_INPLACE_SHIFT": "1"}, lambda: plan.interp_src(table, out=out)),
                ("in place, straddling", {"S3_INPLACE_SHIFT": "0"}, lambda: plan.interp_src(table, out=out)),
                ("pitched compacted copy", {}, lambda: plan.interp(w, rows, out=out)),
                ("pitched, shift kernel", {"S3_INPLACE_SHIFT": "2"}, lambda: plan.interp(w, rows, out=out))]
    if os.environ.get("AB_SHIFT_MID") == "1":          # mid-length rows: the shift kernel instead of the persistent kernel
        variants.append(("in place, shift kernel", {"S3_INPLACE_SHIFT": "1", "S3_SHIFT_MIN_CHUNKS": "1"}, lambda: plan.interp_src(table, out=out)))
        variants.append(("in place, persistent kernel", {"S3_INPLACE_SHIFT": "1", "S3_SHIFT_MIN_CHUNKS": "99"}, lambda: plan.interp_src(table, out=out)))
    if os.environ.get("AB_DENSE_COMPACT") == "1":       # the referenced rows only, dense (pitch = row length), Hilbert order
        dense = rows.contiguous()
        variants.append(("dense compacted copy", {}, lambda: plan.interp(w, dense, out=out)))
    direct = hipops.interp(w, idx, table)
    for name, env, fn in variants:
        os.environ.update(env)
        out.zero_()
        fn()
        pt.cuda.synchronize()
        print(f"T={t}: {name}: same bits as the direct kernel: {bool(pt.equal(out, direct))}", flush=True)
        for kk in env:
            del os.environ[kk]
    del direct
    times = {v[0]: [] for v in variants}
    for r in range(rounds + 1):
        for name, env, fn in variants:
            os.environ.update(env)
            e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); pt.cuda.synchronize()
            for kk in env:
                del os.environ[kk]
            if r:
                times[name].append(e0.elapsed_time(e1) / reps)
    b_alg = int(used.numel()) * t * 4 + nc * t * 8 + nc * k * 12
    for name, tt in times.items():
        med = statistics.median(tt)
        print(f"T={t}: {name:26s} median {med:.4f} ms  min {min(tt):.4f}  max {max(tt):.4f}   frac of 8 TB/s {b_alg / (med * 1e-3) / 8e12:.3f}", flush=True)
    del table, rows, out
