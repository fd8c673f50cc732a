"""dev helper: steady-state time per batch of ExportData._fit_data on the bench grid (host batches back to back), with the
download of a batch completing behind the next batch's upload (default) or waited for in the call (S3_EXPORT_DEFER=0)
    python tools/e2e_probe.py T [reps]"""
import os, sys, time, types, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.export import ExportData
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
t = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
n, nc = len(x), len(centers)
s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=None, levels=None,
                          metric=pt.zeros(n, dtype=pt.float64), size_initial_cell=1.0, save_path=".", save_name="b", grid_name="g")
ex = ExportData(s, write_times=[str(i) for i in range(100000)], n_neighbors=26)
coords = pt.from_numpy(x)
data = pt.empty((n, 1, t), dtype=pt.float32).normal_()
for _ in range(2):
    ex._fit_data(coords, data, "f", 10 ** 9)
pt.cuda.synchronize()
for mode in ("1", "0", "1", "0"):
    os.environ["S3_EXPORT_DEFER"] = mode
    pt.cuda.synchronize()
    t0 = time.perf_counter()
    calls = []
    for _ in range(reps):
        c0 = time.perf_counter()
        ex._fit_data(coords, data, "f", 10 ** 9)
        calls.append((time.perf_counter() - c0) * 1e3)
    pt.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print(f"T={t} defer={mode}: {dt:.1f} ms per batch ({nc * t / dt / 1e6:.2f} G cell*snapshots/s); host time per call " +
          " ".join(f"{c:.1f}" for c in calls), flush=True)
