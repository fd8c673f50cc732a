"""dev helper (GPU box, probe build -DS3_PROBE_STAMPS of interp_plan.hip, see tools/stream_phases.sh): mean duration of the phases of
one step of the persistent kernel, from s_memrealtime stamps summed per wavefront
    python tools/stream_phases.py T [T ...]"""
import sys, ctypes as C, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops, _lib
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
ts = [int(a) for a in sys.argv[1:]] or [25, 32, 100]
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
k = 26
knn = hipops.KnnIndex(x); idx, dist = knn.query(centers, k); knn.close()
w = hipops.idw_weights(dist)
plan = hipops.InterpPlan(idx, len(x), centers)
lib = _lib.hip_lib()
lib.s3_probe_read.argtypes = [C.c_void_p, C.c_int]
names = ["wait for the lines + LDS stores", "barrier 1", "row ids + issue of the next step's loads", "accumulate", "output stores", "barrier 2"]
for t in ts:
    data = hipops.padded_rows(len(x), t, pt.float32, "cuda", 0)
    data.normal_()
    out = pt.empty((len(centers), t), dtype=pt.float64, device="cuda")
    for _ in range(3):
        plan.interp(w, data, out=out)
    e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    e0.record(); plan.interp(w, data, out=out); e1.record(); pt.cuda.synchronize()
    buf = np.zeros(1024 * 4 * 8, dtype=np.int64)
    assert lib.s3_probe_read(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    s = buf.reshape(1024, 4, 8)
    used = s[:, :, 6] > 0
    steps = s[:, :, 6][used].astype(float)
    print(f"T={t}: launch {e0.elapsed_time(e1) * 1e3:.1f} us; workgroups {int(used[:, 0].sum())}, steps per workgroup {steps.mean():.1f}; per step (mean over wavefronts, us):")
    total = 0.0
    for i, nme in enumerate(names):
        d = (s[:, :, i][used] / steps).mean() * 0.01          # 100-MHz ticks
        total += d
        print(f"    {nme:45s} {d:6.2f}")
    print(f"    {'sum':45s} {total:6.2f}   (x steps = {total * steps.mean():.1f} us)")
