"""
S^3 on a synthetic stand-in for the reference's cylinder3D_Re3900 example (reference
examples/s3_for_cylinder3D_Re3900.py:84-150; the OpenFOAM data set itself is not redistributable): 3-D channel with a
cylinder, the field arrives in batches of snapshots, the metric is computed batch-wise before the grid is generated,
the fields are exported batch by batch afterwards.

    python examples/s3_for_synthetic_cylinder3D.py [n_points] [save_path]

Differences to a script written against the reference: the import lines, and the metric (temporal standard deviation)
comes from ``metrics.temporal_moments`` -- one streaming pass per batch on the GPU, merged across batches -- instead of
torch on the CPU.  Needs an MI355X; the HDF5/XDMF files are written by the package's own sink (libs3h5.so).
"""
import sys
from os.path import abspath, dirname, join
from time import time

import torch as pt

sys.path.insert(0, dirname(dirname(abspath(__file__))))
from sparsespatialsampling_amd import metrics                                           # noqa: E402
from sparsespatialsampling_amd.export import ExportData                                 # noqa: E402
from sparsespatialsampling_amd.geometry import CubeGeometry, CylinderGeometry3D         # noqa: E402
from sparsespatialsampling_amd.sparse_spatial_sampling import SparseSpatialSampling     # noqa: E402


def snapshots(coord: pt.Tensor, t0: int, t1: int) -> pt.Tensor:
    """synthetic pressure field [N, 1, t1 - t0]: a wake that travels downstream of the cylinder"""
    x, y, z = coord[:, 0:1], coord[:, 1:2], coord[:, 2:3]
    t = pt.arange(t0, t1, dtype=pt.float64)[None, :]
    wake = pt.exp(-((y - 1.0) / 0.15) ** 2) * pt.exp(-0.8 * (x - 0.8).clamp(min=0)) * (x > 0.8)
    p = wake * pt.sin(2 * pt.pi * (x - 0.8) / 0.5 - 2 * pt.pi * t / 40) * (1 + 0.1 * pt.cos(20 * z))
    return (p + 1e-3 * pt.randn(p.shape)).float().unsqueeze(1)


if __name__ == "__main__":
    n_points = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    save_path = sys.argv[2] if len(sys.argv) > 2 else join("run", "cylinder3D_synthetic")
    n_snapshots, batch = 200, 25
    bounds = [[0.0, 0.0, 0.0], [2.4, 2.0, 0.314]]
    cylinder = [[(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05]

    pt.manual_seed(0)
    coord = pt.rand(n_points, 3) * pt.tensor(bounds[1])
    coord = coord[((coord[:, :2] - pt.tensor([0.8, 1.0])) ** 2).sum(1) > cylinder[1] ** 2]

    # metric = temporal standard deviation of the field, accumulated over the batches (Chan's update of count / mean / M2)
    t_start = time()
    count, mean, m2 = 0, None, None
    for t0 in range(0, n_snapshots, batch):
        b_mean, b_std = metrics.temporal_moments(snapshots(coord, t0, t0 + batch).squeeze(1), unbiased=False)
        b_m2 = b_std ** 2 * batch
        if mean is None:
            count, mean, m2 = batch, b_mean, b_m2
        else:
            delta = b_mean - mean
            total = count + batch
            mean, m2, count = mean + delta * batch / total, m2 + b_m2 + delta ** 2 * count * batch / total, total
    metric = (m2 / (count - 1)).sqrt()
    print(f"metric from {n_snapshots} snapshots in {time() - t_start:.2f} s")

    domain = CubeGeometry("domain", True, bounds[0], bounds[1])
    body = CylinderGeometry3D("cylinder", False, cylinder[0], cylinder[1], refine=True)
    s_cube = SparseSpatialSampling(coord, metric, [domain, body], save_path, "metric_0.75", "cylinder3D", min_metric=0.75)
    t_start = time()
    s_cube.execute_grid_generation()
    print(f"generated {s_cube.centers.shape[0]} cells from {coord.shape[0]} original cells in {time() - t_start:.2f} s")

    export = ExportData(s_cube, write_times=[str(i) for i in range(n_snapshots)])
    t_start = time()
    try:
        for t0 in range(0, n_snapshots, batch):
            export.export(coord, snapshots(coord, t0, t0 + batch), "p", n_snapshots_total=n_snapshots)
        print(f"wrote {join(save_path, 'metric_0.75')}.h5 / .xdmf in {time() - t_start:.2f} s")
    except ImportError as e:                  # neither libs3h5.so nor h5py: show the interpolated field instead
        print(f"{e}; interpolated batch: {tuple(export._interpolated_fields.centers.shape)}")
