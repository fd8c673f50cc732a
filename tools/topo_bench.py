"""dev helper (CPU): throughput of the native topology engine (libs3topo.so) -- uniform refinement of a 3-D root cell to
level L, then random adaptive batches"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from sparsespatialsampling_amd.s_cube import _Topology
L = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t = _Topology(3, 1.0, np.array([0.5, 0.5, 0.5]))
leaves = np.array([0], dtype=np.int64)
t0 = time.perf_counter()
for lvl in range(L):
    first = t.refine(leaves, True)
    leaves = np.arange(first, first + 8 * len(leaves), dtype=np.int64)
t1 = time.perf_counter()
print(f"uniform to level {L}: {t.n_cells} cells in {t1-t0:.3f} s = {(t1-t0)/t.n_cells*1e9:.0f} ns/cell")
rng = np.random.default_rng(0)
leaf = list(leaves)
tot = 0
t2 = time.perf_counter()
pool = leaves
for it in range(20):
    pick = rng.choice(pool, size=min(len(pool), 40000), replace=False)
    ta = time.perf_counter()
    first = t.refine(pick, False)
    t_ref = globals().get("t_ref", 0.0) + time.perf_counter() - ta
    new = np.arange(first, first + 8 * len(pick), dtype=np.int64)
    pool = np.concatenate([np.setdiff1d(pool, pick, assume_unique=True), new])
    tot += len(new)
t3 = time.perf_counter()
print(f"adaptive: {tot} cells in {t_ref:.3f} s = {t_ref/tot*1e9:.0f} ns/cell (wall incl. numpy bookkeeping {t3-t2:.3f} s)")
