"""dev helper: ExportData._fit_data on a device-resident T-snapshot batch of the bench grid, called back to back (as an export loop
does) or with a synchronize after every call -- per-batch time of each
    python tools/fit_calls_probe.py [T]"""
import sys, time, types, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.export import ExportData, _as_float
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine(); centers = tree.all_centers.numpy(); tree.close()
s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=None, levels=None,
                          metric=pt.zeros(len(x), dtype=pt.float64), size_initial_cell=1.0, save_path=".", save_name="bench", grid_name="g")
ex = ExportData(s, write_times=[str(i) for i in range(100000)], n_neighbors=26)
coords = pt.from_numpy(x)
data = pt.empty((len(x), 1, t), dtype=pt.float32, device="cuda").normal_()
for _ in range(2):
    ex._fit_data(coords, data, "f", 10 ** 9)
pt.cuda.synchronize()
for mode in ("sync after every call", "back to back", "sync after every call", "back to back"):
    t0 = time.perf_counter()
    for _ in range(3):
        ex._fit_data(coords, data, "f", 10 ** 9)
        if mode.startswith("sync"):
            pt.cuda.synchronize()
    pt.cuda.synchronize()
    print(f"T={t} {mode}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per batch", flush=True)
