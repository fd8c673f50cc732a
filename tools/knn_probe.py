"""dev helper: time of the KNN kernels on the bench cloud (child-metric batches as refine issues them, KNN cache query)"""
import os, sys, statistics
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import hipops
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
occ = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
knn = hipops.KnnIndex(x, target_occupancy=occ); knn.set_values(metric)
print("target occupancy", occ or "default", "buckets", knn.n_buckets, flush=True)
rng = np.random.default_rng(0)
def timeit(f, reps=5):
    f(); pt.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): f()
        b.record(); pt.cuda.synchronize(); ts.append(a.elapsed_time(b) / reps)
    return statistics.median(ts)
q = pt.from_numpy(rng.random((360_000, 3)) * [2.4, 2.0, 0.314]).cuda()
print(f"idw_predict 360k random queries k=26: {timeit(lambda: knn.predict(q, 26)):.3f} ms", flush=True)
# spatially clustered queries in groups of nine (a parent centre and its eight child centres)
c = rng.random((40_000, 3)) * [2.4, 2.0, 0.314]
d = np.array([[0, 0, 0]] + [[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64) * 0.002
qc = pt.from_numpy((c[:, None, :] + d[None]).reshape(-1, 3)).cuda()
print(f"idw_predict 40k x 9 grouped queries k=26: {timeit(lambda: knn.predict(qc, 26)):.3f} ms", flush=True)
# the same groups visited in a spatially coherent order (what sorting a refine batch by cell position would give)
key = (np.floor(c[:, 2] / 0.314 * 16) * 256 + np.floor(c[:, 1] / 2.0 * 16)) * 256 + np.floor(c[:, 0] / 2.4 * 256)
cs_ = c[np.argsort(key, kind="stable")]
qs = pt.from_numpy((cs_[:, None, :] + d[None]).reshape(-1, 3)).cuda()
print(f"idw_predict 40k x 9 grouped queries, cells sorted by position: {timeit(lambda: knn.predict(qs, 26)):.3f} ms", flush=True)
qq = pt.from_numpy(rng.random((461_130, 3)) * [2.4, 2.0, 0.314]).cuda()
print(f"knn query 461k k=26: {timeit(lambda: knn.query(qq, 26)):.3f} ms", flush=True)
