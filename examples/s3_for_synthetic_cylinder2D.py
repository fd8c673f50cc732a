"""
End-to-end S^3 run on a synthetic stand-in for the reference's cylinder2D_Re100 example
(reference examples/s3_for_cylinder2D_Re100.py:33-73; the OpenFOAM data set itself is not redistributable):

    python examples/s3_for_synthetic_cylinder2D.py [save_path]

Only the import lines differ from a script written against the reference.  Needs an MI355X; the HDF5/XDMF files are
written by the package's own sink (libs3h5.so on the HDF5 C library; h5py is used when that library is missing).
"""
import sys
from os.path import abspath, dirname, join

import torch as pt

sys.path.insert(0, dirname(dirname(abspath(__file__))))
from sparsespatialsampling_amd.export import ExportData                                 # noqa: E402
from sparsespatialsampling_amd.geometry import CubeGeometry, SphereGeometry             # noqa: E402
from sparsespatialsampling_amd.metrics import temporal_std                              # noqa: E402
from sparsespatialsampling_amd.sparse_spatial_sampling import SparseSpatialSampling     # noqa: E402

if __name__ == "__main__":
    save_path = sys.argv[1] if len(sys.argv) > 1 else join("run", "cylinder2D_synthetic")
    min_metric = 0.75
    save_name = "metric_{:.2f}".format(min_metric)

    # synthetic "CFD" data: 14 000 cell centres in the channel minus the cylinder, 400 snapshots of a travelling wake
    bounds = [[0, 0], [2.2, 0.41]]
    cylinder = [[0.2, 0.2], 0.05]
    pt.manual_seed(0)
    coord = pt.rand(14000, 2) * pt.tensor(bounds[1])
    coord = coord[(coord - pt.tensor(cylinder[0])).norm(dim=1) > cylinder[1]]
    x, y, t = coord[:, 0:1], coord[:, 1:2], pt.arange(400.0)[None, :]
    wake = pt.exp(-((y - 0.2) / 0.08) ** 2) * pt.exp(-(x - 0.2).clamp(min=0)) * (x > 0.2)
    field = (wake * pt.sin(2 * pt.pi * (x - 0.2) / 0.4 - 2 * pt.pi * t / 40) + 1e-3 * pt.randn(len(coord), 400)).float()
    write_times = [str(round(0.01 * i, 2)) for i in range(400)]

    domain = CubeGeometry("domain", True, bounds[0], bounds[1])
    geometry = SphereGeometry("cylinder", False, cylinder[0], cylinder[1], refine=True, min_refinement_level=9)

    # metric = pt.std(field, dim=1) of the reference's script (OAT15 example, line 91), one streaming pass on the GPU
    s_cube = SparseSpatialSampling(coord, temporal_std(field), [domain, geometry], save_path, save_name,
                                   "cylinder2D", min_metric=min_metric)
    s_cube.execute_grid_generation()
    print(f"generated {s_cube.centers.shape[0]} cells from {coord.shape[0]} original cells")

    export = ExportData(s_cube, write_times=write_times)
    try:
        export.export(coord, field.unsqueeze(1), "p")
        print(f"wrote {join(save_path, save_name)}.h5 / .xdmf")
        # weighted SVD of the exported field (reference utils.write_svd_s_cube_to_file): modes next to the data
        from sparsespatialsampling_amd.svd import write_svd_s_cube_to_file
        write_svd_s_cube_to_file("p", save_path, save_name, export.new_file, n_modes=10, rank=20)
        print(f"wrote {join(save_path, save_name)}_p_svd.h5 / .xdmf")
    except ImportError as e:                  # neither libs3h5.so nor h5py: show the interpolated field instead
        print(f"{e}; interpolated field: {tuple(export._interpolated_fields.centers.shape)}")
