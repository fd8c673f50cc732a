"""dev helper: s3_child_gain_reuse on synthetic batches of new cells of one level each (cylinder3D bench cloud): time with the
wavefront-per-cell kernel on / off (S3_KNN_COOP), results bit-equal.
    python tools/knn_coop_probe.py [n_cells]"""
import os, sys, logging
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import hipops
logging.getLogger().setLevel(logging.WARNING)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
BOX = os.environ.get("S3_PROBE_CLOUD") == "box5e7"       # the 5e7-point unit box of C4 instead of the cylinder3D cloud
cfg = dict(bench.WORKLOADS["box5e7" if BOX else "cylinder3D_Re3900"])
x, metric = bench.synthetic_box(cfg) if BOX else bench.synthetic_cylinder3d(cfg)
k, dim, nch = 26, 3, 8
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, dim))
knn.set_values(metric)
width = 1.0 if BOX else 2.4
rng = np.random.default_rng(0)
lf = hipops.to_device(np.array([1 / 8 * ((width / 2 ** lv) ** 3) for lv in range(64)]))
print(f"points {len(x)}, buckets {knn.n_buckets}; bucket side ~ {(2.4 * 2.0 * 0.314 / knn.n_buckets) ** (1 / 3):.5f}")


def timed(fn, reps=5):
    fn(); pt.cuda.synchronize()
    a, b = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); pt.cuda.synchronize()
    return a.elapsed_time(b) / reps


for lv in ((6, 7, 8) if BOX else (6, 7, 8, 9, 10)):
    cap = n + 8
    x0 = float(os.environ.get("S3_PROBE_X0", "0.2"))
    c_host = (0.05 + 0.9 * rng.random((cap, 3))) if BOX else np.array([x0, 0.2, 0.02]) + rng.random((cap, 3)) * np.array([2.2 - x0, 1.6, 0.27])
    if os.environ.get("S3_PROBE_SORT") == "1":           # cells in a spatial order (bucket-row major) instead of a random one
        key = np.floor(c_host / (0.0034 if BOX else 0.0084)).astype(np.int64)
        c_host = c_host[np.lexsort((key[:, 0], key[:, 1], key[:, 2]))]
    center = pt.from_numpy(np.ascontiguousarray(c_host)).cuda()
    level = pt.full((cap,), lv, dtype=pt.int32, device="cuda")
    metric_d, gain_d = pt.zeros(cap, dtype=pt.float64, device="cuda"), pt.zeros(cap, dtype=pt.float64, device="cuda")
    child = pt.zeros((cap, nch), dtype=pt.float64, device="cuda")
    parents = pt.zeros((n + nch - 1) // nch, dtype=pt.int32, device="cuda")
    scratch = pt.zeros(n * (nch + 1) + 2 + n * nch, dtype=pt.float64, device="cuda")
    out = {}
    for mode in ("1", "0"):
        os.environ["S3_KNN_COOP"] = mode
        run = lambda: hipops.child_gain_reuse(knn, k, center, level, 8, n, width, lf, 1.0, metric_d, gain_d, scratch, parents, 0, child)
        ms = timed(run)
        tail = scratch[n * (nch + 1):].view(pt.int32)
        # second list: what the grouped search left to one wavefront per query; first list (reused): what that left per lane
        left = (int(tail[2 + n * nch]), int(tail[0])) if mode == "1" else (0, 0)
        out[mode] = (ms, child[8:8 + n].clone(), gain_d[8:8 + n].clone(), left)
    same = bool(pt.equal(out["1"][1], out["0"][1]) and pt.equal(out["1"][2], out["0"][2]))
    print(f"level {lv} (cell {width / 2 ** lv:.5f}, quarter {width / 2 ** lv / 4:.5f}): wavefront kernels {out['1'][0]:.3f} ms "
          f"({out['1'][3][0]} of {n * nch} queries to one wavefront each, {out['1'][3][1]} to the per-lane search), per-lane kernel {out['0'][0]:.3f} ms, same bits {same}", flush=True)
