"""dev helper: interleaved A/B timing of planned-interpolation variants in ONE process (variants differ by environment
switches read at plan creation and/or by the row pitch of the snapshot matrix); prints median / min per variant.
    python tools/ab_plan.py "name:ENV=VAL,ENV2=VAL;name2:;name3:PITCH_EXTRA=1"  [rounds] [launches per round]"""
import os, sys, logging, statistics
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
spec = sys.argv[1] if len(sys.argv) > 1 else "base:"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
tree.refine()
centers = tree.all_centers.numpy()
k, T = 26, int(os.environ.get('AB_T', '1000'))
idx, dist = hipops.KnnIndex(x).query(centers, k)
w = hipops.idw_weights(dist)
nc = len(centers)
out = pt.empty((nc, T), dtype=pt.float64, device="cuda")
variants, buffers = [], {}
for item in spec.split(";"):
    name, _, envs = item.partition(":")
    env = dict(e.split("=") for e in envs.split(",") if e)
    extra = int(env.pop("PITCH_EXTRA", "0")), env.pop("BUF", "")       # BUF=<tag>: a separate buffer of the same pitch
    for kk, v in env.items():
        os.environ[kk] = v
    plan = hipops.InterpPlan(idx, len(x), centers, tile_cells=int(env.get("S3_TILE_CELLS", "0")))
    for kk in env:
        del os.environ[kk]
    if extra not in buffers:
        buffers[extra] = hipops.padded_rows(len(x), T, pt.float32, "cuda", extra[0])
        buffers[extra].normal_(generator=pt.Generator(device="cuda").manual_seed(1))
    variants.append((name, plan, buffers[extra], dict(env)))
    print(f"{name}: tiles {plan.n_tiles} staged rows {plan.total_rows} pitch {buffers[extra].stride(0) * 4} B", flush=True)
times = {v[0]: [] for v in variants}
for r in range(rounds + 1):
    for name, plan, data, lenv in variants:
        os.environ.update(lenv)          # switches read at launch time
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        plan.interp(w, data, out=out)
        e0.record()
        for _ in range(reps):
            plan.interp(w, data, out=out)
        e1.record(); pt.cuda.synchronize()
        for kk in lenv:
            del os.environ[kk]
        if r:
            times[name].append(e0.elapsed_time(e1) / reps)
for n, t in times.items():
    print(f"{n:24s} median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}")
