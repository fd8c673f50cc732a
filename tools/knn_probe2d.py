"""dev helper: KNN time vs bucket occupancy on a 2-D cloud (k = 8)"""
import os, sys, statistics
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsespatialsampling_amd import hipops
rng = np.random.default_rng(0)
x = rng.random((3_000_000, 2)); y = rng.random(3_000_000)
q = pt.from_numpy(rng.random((500_000, 2))).cuda()
for occ in (0.6, 0.8, 1.0, 1.2, 1.5, 2.0, 3.0, 5.0):
    knn = hipops.KnnIndex(x, occ); knn.set_values(y)
    knn.predict(q, 8); pt.cuda.synchronize(); ts = []
    for _ in range(5):
        a, b = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): knn.predict(q, 8)
        b.record(); pt.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    print(f"occupancy {occ}: {statistics.median(ts):.3f} ms", flush=True)
    knn.close()
print("rule:", hipops.knn_occupancy(8, 2))
