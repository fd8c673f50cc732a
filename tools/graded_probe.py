"""dev helper: KNN build/query time on a strongly graded (boundary-layer like) point cloud vs a uniform one"""
import sys, time
import numpy as np, torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import hipops
rng = np.random.default_rng(0)
for d, k, n in ((2, 8, 300_000), (3, 26, 2_000_000)):
    for kind in ("uniform", "graded"):
        if kind == "uniform":
            x = rng.random((n, d)) * 2 - 1
        else:
            r = 10 ** rng.uniform(-5, 0, n)                       # log-uniform wall distance
            v = rng.standard_normal((n, d)); v /= np.linalg.norm(v, axis=1, keepdims=True)
            x = v * (0.1 + r)[:, None]                            # shell of radius 0.1 with graded layers around it
        q = x[rng.integers(0, n, 200_000)] + 1e-7                 # queries distributed like the points
        pt.cuda.synchronize(); t0 = time.perf_counter()
        knn = hipops.KnnIndex(x); pt.cuda.synchronize(); t1 = time.perf_counter()
        idx, dist = knn.query(q, k); pt.cuda.synchronize(); t2 = time.perf_counter()
        print(f"d={d} {kind:8s} n={n}: build {t1-t0:.3f}s  query 2e5 x k={k}: {t2-t1:.3f}s", flush=True)
        knn.close()
