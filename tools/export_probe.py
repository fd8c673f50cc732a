"""dev helper: end-to-end ExportData.export() timing with host tensors (PCIe included), HDF5 sink faked in memory"""
import sys, time, logging, types
import numpy as np, torch as pt
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from sparsespatialsampling_amd.export import ExportData
logging.getLogger().setLevel(logging.WARNING)
n, nc = 5_000_000, 461_130
rng = np.random.default_rng(0)
x = rng.random((n, 3)) * [2.4, 2.0, 0.314]
centers = rng.random((nc, 3)) * [2.4, 2.0, 0.314]
s = types.SimpleNamespace(n_dimensions=3, faces=pt.zeros((nc, 8), dtype=pt.int32), centers=pt.from_numpy(centers),
                          vertices=pt.zeros((8, 3)), levels=pt.ones((nc, 1), dtype=pt.int64), metric=pt.rand(n),
                          size_initial_cell=2.4, save_path="/tmp", save_name="probe", grid_name="g")
for t in (25, 200):
    ex = ExportData(s, write_times=[str(i) for i in range(3 * t)])
    coords = pt.from_numpy(x)
    for b in range(3):
        data = pt.randn((n, 1, t), dtype=pt.float32)
        pt.cuda.synchronize(); t0 = time.perf_counter()
        ex._fit_data(coords, data, "p", 3 * t)
        pt.cuda.synchronize(); t1 = time.perf_counter()
        ex._field_name = "p"; ex._write_data_to_hdf5()
        t2 = time.perf_counter()
        print(f"T={t} batch {b}: fit (H2D + kernel + D2H) {t1-t0:.3f}s  sink(fake) {t2-t1:.3f}s  "
              f"-> {nc*t/(t1-t0)/1e6:.1f} M cell*snap/s end-to-end", flush=True)
