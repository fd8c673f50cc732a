"""dev helper: ExportData.export() with the HDF5 file at its end (bench.export_to_file), for the writer's settings given in the
environment (S3H5_WRITE_THREADS, S3H5_WRITE_MODE)
    python tools/file_probe.py T n_batches [directory]"""
import os, sys, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
t = int(sys.argv[1]) if len(sys.argv) > 1 else 25
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if len(sys.argv) > 3:
    import tempfile
    tempfile.tempdir = sys.argv[3]
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
out = (tree.all_centers, tree.all_nodes, tree.face_ids, tree.all_levels, float(tree.width))
tree.close()
for rep in range(2):
    r = bench.export_to_file(x, metric, out, 26, t=t, n_batches=nb)
    print(f"T={t} threads={os.environ.get('S3H5_WRITE_THREADS', '6')} mode={os.environ.get('S3H5_WRITE_MODE', 'pwrite')} dir={r['directory']}: "
          f"steady {r['ms_per_export_call_steady']:.2f} ms/call = {r['Gcells_snapshots_per_s_steady']:.2f} G/s; total {r['total_s']:.3f} s, "
          f"file {r['file_MB_per_s'] / 1e3:.2f} GB/s; calls {r['ms_per_export_call']}", flush=True)
