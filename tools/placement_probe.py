"""dev probe: does the launch time of the in-place headline kernel depend on WHICH allocation holds the dense batch?  Four 20-GB tables
allocated one after the other (all kept), the same plan launched on each, interleaved in one process.  A spread between the tables =
physical placement / page-table fragments of the allocation matter; none = the box's state is global.
    python tools/placement_probe.py [n_tables]"""
import os, sys, logging, statistics
import numpy as np, torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
n_tables = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
k, n, t = 26, len(x), 1000
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
idx, dist = knn.query(centers, k)
w = hipops.idw_weights(dist)
knn.close()
used, remap = hipops.referenced_rows([idx], n, coords=x)
hipops.remap_indices(idx, remap)
plan = hipops.InterpPlan(idx, int(used.numel()), centers)
plan.set_weights(w)
plan.set_source_ids(used.contiguous(), n)
gen = pt.Generator(device="cuda").manual_seed(1)
tables = []
for i in range(n_tables):
    tb = pt.empty((n, t), dtype=pt.float32, device="cuda")
    tb.normal_(generator=gen)
    tables.append(tb)
    print(f"table {i}: data_ptr {tb.data_ptr():#x}", flush=True)
outs = [pt.empty((len(centers), t), dtype=pt.float64, device="cuda") for _ in range(2)]
times = {(i, j): [] for i in range(n_tables) for j in range(2)}
for r in range(6):
    for i, tb in enumerate(tables):
        for j, out in enumerate(outs):
            e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
            plan.interp_src(tb, out=out)
            e0.record()
            for _ in range(8):
                plan.interp_src(tb, out=out)
            e1.record(); pt.cuda.synchronize()
            if r:
                times[(i, j)].append(e0.elapsed_time(e1) / 8)
b_alg = int(used.numel()) * t * 4 + len(centers) * t * 8 + len(centers) * k * 12
for (i, j), v in times.items():
    med = statistics.median(v)
    print(f"table {i}, out {j}: median {med:.4f} ms  min {min(v):.4f}  max {max(v):.4f}  frac {b_alg / (med * 1e-3) / 8e12:.3f}", flush=True)
