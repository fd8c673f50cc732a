"""dev helper: row pitch of the resident batch for short / medium rows -- whole 128-byte lines (hipops.padded_rows) against the
row length rounded up to 16 bytes -- on the bench's own layout (referenced rows of the cylinder3D grid in Hilbert order),
interleaved in one process.    python tools/pitch_probe.py [row_len ...]"""
import os, sys, statistics, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
lens = [int(a) for a in sys.argv[1:]] or [25, 32, 75, 100, 128, 200]
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
k = 26
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
idx, dist = knn.query(centers, k)
knn.close()
w = hipops.idw_weights(dist)
used, remap = hipops.referenced_rows([idx], len(x), coords=x)
n_rows, nc = int(used.numel()), len(centers)
hipops.remap_indices(idx, remap)
plan = hipops.InterpPlan(idx, n_rows, centers)
plan.set_weights(w)
for L in lens:
    tight = (L + 3) // 4 * 4
    bufs = {"lines": hipops.padded_rows(n_rows, L, pt.float32, "cuda"),
            "16B": pt.empty((n_rows, tight), dtype=pt.float32, device="cuda")[:, :L]}
    src = pt.empty((n_rows, L), dtype=pt.float32, device="cuda").normal_()
    for b in bufs.values():
        b.copy_(src)
    out = {name: pt.empty((nc, L), dtype=pt.float64, device="cuda") for name in bufs}
    times = {name: [] for name in bufs}
    for r in range(6):
        for name, b in bufs.items():
            ms = bench.launch_times_ms(lambda: plan.interp(w, b, out=out[name]), 10, 2)
            if r:
                times[name].append(float(np.median(ms)))
    same = bool(pt.equal(out["lines"], out["16B"]))
    b_alg = n_rows * L * 4 + nc * L * 8 + nc * k * 12
    print(f"row_len {L:4d}: pitch {bufs['lines'].stride(0) * 4} B {statistics.median(times['lines']):.4f} ms ({b_alg / statistics.median(times['lines']) / 8e9:.3f}) | "
          f"pitch {tight * 4} B {statistics.median(times['16B']):.4f} ms ({b_alg / statistics.median(times['16B']) / 8e9:.3f}) | same bits {same}", flush=True)
