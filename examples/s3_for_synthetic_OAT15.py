"""
End-to-end S^3 run on a synthetic stand-in for the reference's OAT15 airfoil example (BASELINE config C2;
reference examples/s3_for_OAT15_airfoil.py:71-133; the CFD data set and the airfoil's STL file are not redistributable):

    python examples/s3_for_synthetic_OAT15.py [save_path] [n_snapshots]

Same workflow as the reference's script: metric = std over time of the fields, a rectangular domain plus an airfoil given by
the coordinates of its outline (``GeometryCoordinates2D``, refined), grid generation with a cell budget, export of a scalar
and of a vector field in snapshot batches, weighted SVD of the exported field.  What is synthetic: 3 * 10^5 points (half
uniform, half clustered around an analytic NACA-0012 outline), a buffet-like flow (a shock region oscillating on the suction
side + a wake shedding behind the trailing edge), 2000 snapshots by default.  The grid of this configuration is pinned against
the real reference in tests/test_gpu_refine.py::test_c2_oat15_full_size_matches_reference.  Needs an MI355X.
"""
import sys
from os.path import abspath, dirname, join

import numpy as np
import torch as pt

sys.path.insert(0, dirname(dirname(abspath(__file__))))
from sparsespatialsampling_amd.export import ExportData                                 # noqa: E402
from sparsespatialsampling_amd.geometry import CubeGeometry, GeometryCoordinates2D      # noqa: E402
from sparsespatialsampling_amd.metrics import RunningMoments                            # noqa: E402
from sparsespatialsampling_amd.sparse_spatial_sampling import SparseSpatialSampling     # noqa: E402


def naca0012_outline(n: int = 200, chord: float = 1.0, thickness: float = 0.12) -> np.ndarray:
    """closed outline, trailing edge -> suction side -> leading edge -> pressure side (the order the reference's STL loader
    produces, examples/s3_for_OAT15_airfoil.py:23-68)"""
    xs = 0.5 * (1 - np.cos(np.linspace(0, np.pi, n // 2)))
    yt = 5 * thickness * (0.2969 * np.sqrt(xs) - 0.126 * xs - 0.3516 * xs ** 2 + 0.2843 * xs ** 3 - 0.1036 * xs ** 4)
    upper = np.stack([xs[::-1], yt[::-1]], 1)
    lower = np.stack([xs, -yt], 1)[1:-1]
    return np.concatenate([upper, lower]) * chord


def synthetic_fields(x: np.ndarray, t0: int, t1: int):
    """snapshots t0 .. t1-1: p [N, 1, T], U [N, 2, T] float32"""
    t = np.arange(t0, t1, dtype=np.float64)[None, :]
    xx, yy = x[:, :1], x[:, 1:2]
    shock = np.exp(-((xx - 0.45 - 0.08 * np.sin(2 * np.pi * t / 80.0)) / 0.05) ** 2) * np.exp(-((yy - 0.12) / 0.18) ** 2)
    wake = np.where(xx > 1.0, np.exp(-(xx - 1.0) / 0.5), 0.0) * np.exp(-(yy / 0.06) ** 2) * np.sin(2 * np.pi * (t / 25.0 - 3 * xx))
    p = 1.0 + 0.4 * shock + 0.15 * wake
    u = 1.0 - 0.5 * shock + 0.2 * wake
    v = 0.3 * wake * np.cos(2 * np.pi * t / 25.0) + 0.1 * shock
    return pt.from_numpy(p.astype(np.float32)[:, None, :]), pt.from_numpy(np.stack([u, v], 1).astype(np.float32))


if __name__ == "__main__":
    save_path = sys.argv[1] if len(sys.argv) > 1 else join("run", "OAT15_synthetic")
    n_snapshots = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    batch = 250                                        # snapshots per export() call (the fields never exist as a whole)
    save_name = "OAT15_synthetic_n_cells_25000"

    # the "CFD grid": points in [-0.2, 1.2] x [-0.5, 0.5], half of them clustered around the airfoil
    rng = np.random.default_rng(1)
    outline = naca0012_outline()
    far = rng.random((150000, 2)) * [1.4, 1.0] + [-0.2, -0.5]
    near = outline[rng.integers(0, len(outline), 150000)] + 0.02 * rng.standard_normal((150000, 2))
    xz = np.concatenate([far, near])
    xz = np.ascontiguousarray(xz[(xz[:, 0] >= -0.2) & (xz[:, 0] <= 1.2) & (xz[:, 1] >= -0.5) & (xz[:, 1] <= 0.5)])
    write_times = [str(round(1e-3 * i, 3)) for i in range(n_snapshots)]

    # metric = std_t(p) + std_t(|U|) (reference: pt.std(field, dim=1), line 91), one streaming pass per field on the GPU, in batches
    mom_p, mom_u = RunningMoments(), RunningMoments()
    for t0 in range(0, n_snapshots, batch):
        p, u = synthetic_fields(xz, t0, min(n_snapshots, t0 + batch))
        mom_p.update(p.cuda())
        mom_u.update(u.cuda().norm(dim=1, keepdim=True))
    metric = (mom_p.std() + mom_u.std()).reshape(-1).cpu()

    bounds = [[-0.2, -0.5], [1.2, 0.5]]
    geometry = [CubeGeometry("domain", True, bounds[0], bounds[1]),
                GeometryCoordinates2D("OAT15", False, outline, refine=True)]

    s_cube = SparseSpatialSampling(pt.from_numpy(xz), metric, geometry, save_path, save_name, "OAT15", uniform_levels=5,
                                   n_cells_max=25000, max_delta_level=False)
    s_cube.execute_grid_generation()
    print(f"generated {s_cube.centers.shape[0]} cells from {xz.shape[0]} original cells")

    export = ExportData(s_cube, write_times=write_times)
    try:
        for name, pick in (("p", 0), ("U", 1)):
            for t0 in range(0, n_snapshots, batch):
                fields = synthetic_fields(xz, t0, min(n_snapshots, t0 + batch))
                export.export(pt.from_numpy(xz), fields[pick], name, n_snapshots_total=n_snapshots)
        print(f"wrote {join(save_path, save_name)}.h5 / .xdmf")
        from sparsespatialsampling_amd.svd import write_svd_s_cube_to_file
        write_svd_s_cube_to_file("p", save_path, save_name, export.new_file, 50, rank=int(1e5))
        print(f"wrote {join(save_path, save_name)}_p_svd.h5 / .xdmf")
    except ImportError as e:                  # neither libs3h5.so nor h5py: show the interpolated field instead
        print(f"{e}; interpolated field: {tuple(export._interpolated_fields.centers.shape)}")
