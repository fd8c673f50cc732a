"""dev helper: where ExportData._fit_data spends its time on the bench grid -- device-resident or host batch of T snapshots
    python tools/fit_breakdown.py T [cuda|host]"""
import sys, time, types, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.export import ExportData, _as_float
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
t = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
where = sys.argv[2] if len(sys.argv) > 2 else "cuda"
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
data = pt.empty((n, 1, t), dtype=pt.float32, device=where if where == "cuda" else "cpu").normal_()
ex._fit_data(coords, data, "f", 10 ** 9)
ex._fit_data(coords, data, "f", 10 ** 9)


def tick():
    pt.cuda.synchronize(); return time.perf_counter()


for rep in range(3):
    t0 = tick()
    batch, in_place = ex._upload(_as_float(data)); t1 = tick()
    dev = ex._table_centers.apply(batch, True, full_table=in_place); t2 = tick()
    tr = hipops.snapshot_major(dev, 1, t); t3 = tick()
    out = ex._download(dev, 1, t, "centers"); t4 = tick()
    ex._fit_data(coords, data, "f", 10 ** 9); t5 = tick()
    print(f"T={t} {where}: upload {1e3*(t1-t0):6.1f}  interp {1e3*(t2-t1):6.2f}  transpose alone {1e3*(t3-t2):6.2f}  "
          f"_download (transpose + D2H) {1e3*(t4-t3):6.1f}  | whole _fit_data {1e3*(t5-t4):6.1f} ms", flush=True)
