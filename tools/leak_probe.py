"""dev helper: device-memory leak check -- handles created and destroyed in a loop must not grow the memory in use"""
import os, sys
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsespatialsampling_amd import hipops, geometry, metrics
from sparsespatialsampling_amd.s_cube import SamplingTree
import logging; logging.disable(logging.CRITICAL)
rng = np.random.default_rng(0)
x = rng.random((200_000, 3)); y = rng.random(200_000); c = rng.random((20_000, 3))
def used():
    pt.cuda.synchronize(); pt.cuda.empty_cache()
    free, total = pt.cuda.mem_get_info()
    return (total - free) / 2**20
base = None
for it in range(6):
    for _ in range(20):
        knn = hipops.KnnIndex(x, 2.0); knn.set_values(y)
        idx, dist = knn.query(c, 26); w = hipops.idw_weights(dist)
        plan = hipops.InterpPlan(idx, len(x), c)
        data = hipops.padded_rows(len(x), 64, pt.float32, "cuda"); hipops.upload_rows(pt.randn(len(x), 64, dtype=pt.float32), data)
        out = plan.interp(w, data)
        metrics.temporal_std(data)
        plan.close(); knn.close()
        del knn, plan, idx, dist, w, data, out
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(y), [geometry.CubeGeometry("d", True, [0, 0, 0], [1, 1, 1])], uniform_level=3, min_metric=0.5)
    tree.refine(); tree._backend.close(); del tree
    m = used()
    base = m if base is None else base
    print(f"round {it}: {m:.0f} MiB in use (first round {base:.0f})", flush=True)
assert m - base < 64, "device memory grows"
print("no growth")
