"""dev helper: device-resident batch [N, 1, T] fp32 -> ExportData's upload step + neighbour table, read in place vs gathered
into the pitched copy first (S3_EXPORT_INPLACE=0), on the cylinder3D bench grid; point order random (the bench cloud) or
Hilbert-sorted (a mesh whose numbering follows space, S3_PROBE_SORTED=1).
    python tools/inplace_probe.py [T ...]"""
import os, sys, types, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.export import ExportData, _as_float
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
ts = [int(a) for a in sys.argv[1:]] or [25, 75, 100, 256, 1000]
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
if os.environ.get("S3_PROBE_SORTED") == "1":
    order = hipops.spatial_order(x).cpu().numpy()
    x, metric = np.ascontiguousarray(x[order]), np.ascontiguousarray(metric[order])
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
tree.refine()
centers = tree.all_centers.numpy()
tree.close()
s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=None, levels=None,
                          metric=pt.zeros(len(x), dtype=pt.float64), size_initial_cell=1.0, save_path=".", save_name="b", grid_name="g")
ex = ExportData(s, write_times=[str(i) for i in range(100000)], n_neighbors=26)
coords = pt.from_numpy(x)
for t in ts:
    data = pt.empty((len(x), 1, t), dtype=pt.float32, device="cuda").normal_()
    ex._fit_data(coords, data, "f", 10 ** 9)
    res = {}
    for mode in ("1", "0"):
        os.environ["S3_EXPORT_INPLACE"] = mode

        def run():
            batch, in_place = ex._upload(_as_float(data))
            return ex._table_centers.apply(batch, True, full_table=in_place)
        ms = bench.launch_times_ms(run, 10, 2)
        res[mode] = (float(np.median(ms)), run())
    print(f"T={t:5d}: in place {res['1'][0]:.3f} ms, gathered first {res['0'][0]:.3f} ms, same bits {bool(pt.equal(res['1'][1], res['0'][1]))}", flush=True)
    del data, res
