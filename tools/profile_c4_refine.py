"""dev helper: host profile of refine() at C4 scale (5e7 points, n_cells_max 1e7)"""
import cProfile, pstats, sys, time, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import geometry
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
rng = np.random.default_rng(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
x = rng.random((n, 3))
r = np.sqrt(((x - 0.5) ** 2).sum(1))
m = 0.05 + np.exp(-6 * r) * (1 + 0.5 * np.sin(25 * x[:, 0]) * np.cos(17 * x[:, 1]))
geos = [geometry.CubeGeometry("domain", True, [0, 0, 0], [1, 1, 1])]
t0 = time.perf_counter()
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(m), geos, uniform_level=5, n_cells=n // 5)
pt.cuda.synchronize(); t1 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
tree.refine(); pt.cuda.synchronize()
pr.disable(); t2 = time.perf_counter()
print("init %.2f s refine %.2f s" % (t1 - t0, t2 - t1))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
