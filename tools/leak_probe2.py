"""dev helper: which handle type grows the device memory in use (leak isolation)"""
import os, sys
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsespatialsampling_amd import hipops, geometry, metrics
from sparsespatialsampling_amd.s_cube import SamplingTree
import logging; logging.disable(logging.CRITICAL)
rng = np.random.default_rng(0)
x = rng.random((200_000, 3)); y = rng.random(200_000); c = rng.random((20_000, 3))
def used():
    import gc; gc.collect()
    pt.cuda.synchronize(); pt.cuda.empty_cache()
    free, total = pt.cuda.mem_get_info()
    return (total - free) / 2**10
knn0 = hipops.KnnIndex(x, 2.0); idx0, dist0 = knn0.query(c, 26); w0 = hipops.idw_weights(dist0)
def t_knn():
    k = hipops.KnnIndex(x, 2.0); k.set_values(y); k.query(c, 26); k.close()
def t_knn_graded():
    xx = x.copy(); xx[:, 0] = xx[:, 0] ** 6
    k = hipops.KnnIndex(xx, 2.0); k.query(c, 26); k.close()
def t_plan():
    p = hipops.InterpPlan(idx0, len(x), c); p.close()
def t_upload():
    d = hipops.padded_rows(len(x), 64, pt.float32, "cuda"); hipops.upload_rows(pt.randn(len(x), 64, dtype=pt.float32), d); pt.cuda.synchronize()
def t_tree():
    t = SamplingTree(pt.from_numpy(x), pt.from_numpy(y), [geometry.CubeGeometry("d", True, [0, 0, 0], [1, 1, 1])], uniform_level=3, min_metric=0.5)
    t.refine(); t._backend.close()
for name, f, n in (("knn", t_knn, 100), ("knn graded (two levels)", t_knn_graded, 100), ("plan", t_plan, 100), ("upload", t_upload, 50), ("tree", t_tree, 10)):
    f(); a = used()
    for _ in range(n): f()
    b = used()
    print(f"{name}: {(b - a) / n:8.1f} KiB per iteration ({a/1024:.0f} -> {b/1024:.0f} MiB)", flush=True)
