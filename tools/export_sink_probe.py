"""dev helper: ExportData.export() end to end on the bench grid with the real HDF5 sink: per-batch time of interpolation + transport
and of the whole export (background writer included), file size, sink throughput"""
import os, sys, time, types, logging, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry
from sparsespatialsampling_amd.export import ExportData
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
t, n_batches = (int(sys.argv[1]) if len(sys.argv) > 1 else 25), (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw); tree.refine()
d = tempfile.mkdtemp(dir="/tmp")
s = types.SimpleNamespace(n_dimensions=3, faces=tree.face_ids, centers=tree.all_centers, vertices=tree.all_nodes, levels=tree.all_levels,
                          metric=pt.from_numpy(metric), size_initial_cell=tree.width, save_path=d, save_name="probe", grid_name="g")
nc = len(tree.all_centers); tree.close()
ex = ExportData(s, write_times=[str(i) for i in range(t * n_batches)])
coords = pt.from_numpy(x)
batches = [pt.randn((len(x), 1, t), dtype=pt.float32) for _ in range(2)]
t0 = time.perf_counter()
per = []
for b in range(n_batches):
    tb = time.perf_counter()
    ex.export(coords, batches[b % 2], "p", n_snapshots_total=t * n_batches)
    per.append(time.perf_counter() - tb)
total = time.perf_counter() - t0
size = os.path.getsize(os.path.join(d, "probe.h5"))
print(f"{n_batches} batches of {t} snapshots, {nc} cells: per export() call [ms] {[round(1e3 * p, 1) for p in per]}")
print(f"total {total:.2f} s = {nc * t * n_batches / total / 1e9:.3f} G cell*snapshots/s incl. HDF5 ({size / 1e6:.0f} MB file, {size / total / 1e6:.0f} MB/s)")
shutil.rmtree(d)
