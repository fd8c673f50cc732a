"""dev helper: where one ExportData._fit_data call of a 25-snapshot batch spends its time (C3-sized problem)"""
import sys, time, types, logging
import numpy as np, torch as pt
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from sparsespatialsampling_amd import hipops
from sparsespatialsampling_amd.export import ExportData, _as_float
logging.getLogger().setLevel(logging.WARNING)
t = int(sys.argv[1]) if len(sys.argv) > 1 else 25
if len(sys.argv) > 2 and sys.argv[2] == "grid":          # the bench workload's generated grid (references 49 % of the points)
    import bench
    from sparsespatialsampling_amd import geometry
    from sparsespatialsampling_amd.s_cube import SamplingTree
    cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
    x, metric = bench.synthetic_cylinder3d(cfg)
    geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
            geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
    tree.refine()
    centers = tree.all_centers.numpy()
    n, nc = len(x), len(centers)
else:                                                    # random target points: nearly every source point is referenced
    n, nc = 5_000_000, 461_130
    rng = np.random.default_rng(0)
    x = rng.random((n, 3)) * [2.4, 2.0, 0.314]
    centers = rng.random((nc, 3)) * [2.4, 2.0, 0.314]
s = types.SimpleNamespace(n_dimensions=3, faces=pt.zeros((nc, 8), dtype=pt.int32), centers=pt.from_numpy(centers),
                          vertices=pt.zeros((8, 3)), levels=pt.ones((nc, 1), dtype=pt.int64), metric=pt.rand(n),
                          size_initial_cell=2.4, save_path="/tmp", save_name="probe", grid_name="g")
ex = ExportData(s, write_times=[str(i) for i in range(4 * t)])
coords = pt.from_numpy(x)
ex._fit_data(coords, pt.randn((n, 1, t), dtype=pt.float32), "p", 4 * t)          # builds the caches
print("referenced source rows:", "all" if ex._used_rows is None else f"{ex._used_rows.numel()} of {n}")
def tick():
    pt.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    data = pt.randn((n, 1, t), dtype=pt.float32)
    t0 = tick()
    batch, in_place = ex._upload(_as_float(data)); t1 = tick()
    dev = ex._table_centers.apply(batch, full_table=in_place); t2 = tick()
    out = ex._download(dev, 1, t, "centers"); t3 = tick()
    print(f"T={t}: upload {1e3*(t1-t0):6.1f}  interp {1e3*(t2-t1):5.2f}  transpose + download {1e3*(t3-t2):6.1f} ms   "
          f"(total {1e3*(t3-t0):.1f})", flush=True)
