"""
SamplingTree.refine() on the MI355X (HIP backend) against the reference's golden vectors and the oracle backend.
GPU only.
"""
import os

import numpy as np
import pytest
import torch as pt

pytestmark = pytest.mark.gpu

from inputs import cloud, refine_inputs, sha, wake_metric                      # noqa: E402
from tests.test_tree_host_logic import check_outputs_against_golden, check_tree_against_golden, load  # noqa: E402


@pytest.mark.parametrize("d", [2, 3])
def test_uniform_tables_gpu(d):
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    z = load(f"uniform_{d}d")
    tree = s_cube.SamplingTree(pt.from_numpy(z["x"]), pt.from_numpy(z["y"]), uniform_level=3,
                               geometry_obj=[geometry.CubeGeometry("domain", True, [0] * d, [10] * d)])
    assert tree._backend.name == "hip"
    tree._refine_uniform()
    check_tree_against_golden(tree, z)


@pytest.mark.parametrize("name", ["refine_2d_metric", "refine_2d_ncells", "refine_2d_delta", "refine_3d_metric",
                                  "refine_3d_delta", "refine_3d_ncells_cone", "refine_2d_triangle", "refine_3d_polytopes",
                                  "refine_2d_polygon"])
def test_refine_matches_reference_gpu(name):
    """cell ids / levels / centres / faces / vertices / per-cell metric + gain: bit-exact vs the real reference"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    z = load(name)
    x, y, geos, kw = refine_inputs(name, geometry)
    assert sha(x, y) == str(z["input_sha"])
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
    assert tree._backend.name == "hip"
    tree.refine()
    check_tree_against_golden(tree, z)
    check_outputs_against_golden(tree, z)


def test_refine_larger_vs_oracle_backend(monkeypatch):
    """a case well beyond the golden sizes (3-D, 60k points): HIP backend vs oracle backend on the same inputs"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    from tests.oracle_backend import OracleTreeBackend
    x = cloud(123, 60000, [0, 0, 0], [2.4, 2.0, 0.4])
    y = wake_metric(x, [0.8, 1.0, 0.0], decay=2.5)

    def geos():
        return [geometry.CubeGeometry("domain", True, [0, 0, 0], [2.4, 2.0, 0.4]),
                geometry.CylinderGeometry3D("cyl", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.12, refine=True)]

    kw = dict(uniform_level=4, min_metric=0.5)
    t_hip = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos(), **kw)
    t_hip.refine()
    monkeypatch.setattr(s_cube, "_make_backend", lambda v, t, k: OracleTreeBackend(v, t, k))
    t_orc = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos(), **kw)
    t_orc.refine()
    assert pt.equal(t_hip.all_centers, t_orc.all_centers) and pt.equal(t_hip.all_levels, t_orc.all_levels)
    assert pt.equal(t_hip.face_ids, t_orc.face_ids) and pt.equal(t_hip.all_nodes, t_orc.all_nodes)
    vh, vo = t_hip._cell_values(), t_orc._cell_values()
    assert np.array_equal(vh["metric"], vo["metric"]) and np.array_equal(vh["gain"], vo["gain"])
    np.testing.assert_allclose(t_hip._metric, t_orc._metric, rtol=1e-12)
    assert t_hip._n_cells_log == t_orc._n_cells_log
    # geometric sanity of the output: every centre is the mean of its vertices
    c = t_hip.all_nodes[t_hip.face_ids.long()].mean(1)
    assert pt.allclose(c, t_hip.all_centers, rtol=0, atol=1e-12)


def test_export_pipeline_gpu(tmp_path):
    """SparseSpatialSampling -> pickled s_cube object -> ExportData KNN cache + interpolation (HDF5 sink stubbed)"""
    from sparsespatialsampling_amd import geometry
    from sparsespatialsampling_amd.export import ExportData
    from sparsespatialsampling_amd.sparse_spatial_sampling import SparseSpatialSampling
    from oracle import s3_oracle as orc
    x, y, geos, kw = refine_inputs("refine_2d_metric", geometry)
    s3 = SparseSpatialSampling(pt.from_numpy(x), pt.from_numpy(y), geos, str(tmp_path), "case", uniform_levels=4,
                               min_metric=0.6)
    s3.execute_grid_generation()
    z = load("refine_2d_metric")
    assert np.array_equal(s3.centers.numpy(), z["all_centers"]) and np.array_equal(s3.levels.numpy(), z["all_levels"])
    loaded = pt.load(os.path.join(str(tmp_path), "s_cube_case.pt"), weights_only=False)
    assert pt.equal(loaded.centers, s3.centers)

    ex = ExportData(loaded, write_times=[str(i) for i in range(6)])
    rng = np.random.default_rng(5)
    data = rng.standard_normal((len(x), 2, 6)).astype(np.float32)
    ex._fit_data(pt.from_numpy(x), pt.from_numpy(data), "U", None)
    idx_o, dist_o = orc.knn(x, z["all_centers"], 8)
    ref = orc.interp(orc.idw_weights(dist_o), idx_o, data)
    got = ex._interpolated_fields.centers.numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    table = ex._table_centers.idx.cpu().numpy()
    if ex._used_rows is not None:                               # the table addresses the referenced rows by position
        table = ex._used_rows.cpu().numpy()[table]
    assert np.array_equal(table, idx_o)
    ref_metric = orc.interp(orc.idw_weights(dist_o), idx_o, y)
    assert np.abs(ex._metric.numpy() - ref_metric).max() <= 1e-13 * np.abs(ref_metric).max()


def test_export_to_real_hdf5_and_back_gpu(tmp_path):
    """``ExportData.export()`` end to end on the GPU into a REAL HDF5 file (native sink, background writer, two download
    buffers): a scalar field in batches of 7 + 7 + 3 snapshots and a vector field at once, interpolation at the vertices
    too; read back with ``Dataloader`` and compared with the oracle; layout as the reference's loader expects
    (tests/test_s_cube_dataloader.py:40-57: write times, field names, shapes) and as export.py:283-299 writes it (one
    dataset ``data/<t>/<field>_center`` per write time; scalars squeezed)."""
    from sparsespatialsampling_amd import geometry, h5io
    from sparsespatialsampling_amd.data import Dataloader
    from sparsespatialsampling_amd.export import ExportData
    from sparsespatialsampling_amd.sparse_spatial_sampling import SparseSpatialSampling
    from oracle import s3_oracle as orc
    if h5io.native_lib() is None:
        pytest.importorskip("h5py")
    x, y, geos, kw = refine_inputs("refine_2d_metric", geometry)
    s3 = SparseSpatialSampling(pt.from_numpy(x), pt.from_numpy(y), geos, str(tmp_path), "case", uniform_levels=4, min_metric=0.6)
    s3.execute_grid_generation()
    n_t = 17
    times = [f"{0.05 * i:.2f}" for i in range(n_t)]
    rng = np.random.default_rng(9)
    p = rng.standard_normal((len(x), 1, n_t)).astype(np.float32)
    u = rng.standard_normal((len(x), 2, n_t)).astype(np.float32)
    ex = ExportData(s3, write_times=times, interpolate_at_vertices=True)
    for a, b in ((0, 7), (7, 14), (14, 17)):
        ex.export(pt.from_numpy(x), pt.from_numpy(p[:, :, a:b]), "p", n_snapshots_total=n_t)
    ex.export(pt.from_numpy(x), pt.from_numpy(u), "U")

    ld = Dataloader(str(tmp_path), "case.h5", dtype=pt.float64)
    nc, nv = len(s3.centers), len(s3.vertices)
    assert ld.write_times == sorted(times) and ld.field_names[times[3]] == ["U", "p"]
    assert ld.vertices.shape == (nc, 2) and ld.nodes.shape == (nv, 2) and ld.faces.shape == (nc, 4) and ld.faces.dtype == pt.int32
    assert pt.equal(ld.vertices, s3.centers) and pt.equal(ld.nodes, s3.vertices) and pt.equal(ld.faces, s3.faces)
    assert pt.equal(ld.levels, s3.levels.squeeze()) and ld.weights.shape == ld.levels.shape
    idx_c, dist_c = orc.knn(x, s3.centers.numpy(), 8)
    idx_v, dist_v = orc.knn(x, s3.vertices.numpy(), 8)
    w_c, w_v = orc.idw_weights(dist_c), orc.idw_weights(dist_v)
    got_p, got_u = ld.load_snapshot("p", times), ld.load_snapshot("U", times)
    assert got_p.shape == (nc, n_t) and got_u.shape == (nc, 2, n_t)
    ref_p, ref_u = orc.interp(w_c, idx_c, p), orc.interp(w_c, idx_c, u)
    assert np.abs(got_p.numpy() - ref_p[:, 0, :]).max() <= 1e-13 * np.abs(ref_p).max()
    assert np.abs(got_u.numpy() - ref_u).max() <= 1e-13 * np.abs(ref_u).max()
    np.testing.assert_allclose(ld.metric.numpy(), orc.interp(w_c, idx_c, y), rtol=1e-13)
    with h5io.open_h5(os.path.join(str(tmp_path), "case.h5"), "r") as f:       # vertex values and the raw layout
        assert f.keys() == ["constant", "data", "grid"] and f.keys("data/0.10") == ["U_center", "U_vertices", "p_center", "p_vertices"]
        assert f.shape("data/0.10/p_center") == (nc,) and f.shape("data/0.10/U_vertices") == (nv, 2)
        ref_pv = orc.interp(w_v, idx_v, p)
        for i in (0, 6, 7, 16):
            assert np.abs(f.read(f"data/{times[i]}/p_vertices") - ref_pv[:, 0, i]).max() <= 1e-13 * np.abs(ref_pv).max()
    xdmf = open(os.path.join(str(tmp_path), "case.xdmf")).read()
    assert xdmf.count("<Time Value=") == n_t and f'case.h5:/data/{times[-1]}/U_vertices' in xdmf

    # downstream: weighted SVD of the exported fields written next to them (reference utils.py:349-413)
    from sparsespatialsampling_amd import svd
    svd.write_svd_s_cube_to_file(["p", "U"], str(tmp_path), "case", new_file=False, n_modes=3, rank=5)
    for field, comps in (("p", 1), ("U", 2)):
        with h5io.open_h5(os.path.join(str(tmp_path), f"case_{field}_svd.h5"), "r") as f:
            assert f.keys("constant") == ["V", "cell_area", "mode_1", "mode_2", "mode_3", "s"]
            mode, s_val, v_mat, area = f.read("constant/mode_1"), f.read("constant/s"), f.read("constant/V"), f.read("constant/cell_area")
            assert mode.shape == ((nc,) if comps == 1 else (nc, comps)) and s_val.shape == (5,) and v_mat.shape == (n_t, 5)
        data = pt.from_numpy(ref_p[:, 0, :] if comps == 1 else ref_u)
        xw = (data - data.mean(-1, keepdim=True)) * pt.from_numpy(area).sqrt().reshape((-1,) + (1,) * (data.dim() - 1))
        s_ref = pt.linalg.svdvals(xw.reshape(-1, n_t))
        np.testing.assert_allclose(s_val, s_ref[:5].numpy(), rtol=1e-8)
        assert 'Attribute Name="mode_2"' in open(os.path.join(str(tmp_path), f"case_{field}_svd.xdmf")).read()


def _naca_polygon(n=120, chord=1.0, t=0.12):
    """closed NACA-00xx outline (the OAT15-like body of BASELINE config C2)"""
    xs = 0.5 * (1 - np.cos(np.linspace(0, np.pi, n // 2)))
    yt = 5 * t * (0.2969 * np.sqrt(xs) - 0.126 * xs - 0.3516 * xs ** 2 + 0.2843 * xs ** 3 - 0.1036 * xs ** 4)
    upper = np.stack([xs, yt], 1)
    lower = np.stack([xs[::-1], -yt[::-1]], 1)[1:-1]
    return np.concatenate([upper, lower]) * chord


def test_refine_polygon_body_2d_vs_oracle_backend(monkeypatch):
    """C2-like: 2-D, polygon body (GeometryCoordinates2D, refined), n_cells_max stopping: HIP backend == oracle backend"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    from tests.oracle_backend import OracleTreeBackend
    rng = np.random.default_rng(11)
    poly = _naca_polygon()
    x = np.concatenate([rng.random((30000, 2)) * [1.4, 1.0] + [-0.2, -0.5],
                        poly[rng.integers(0, len(poly), 10000)] + 0.03 * rng.standard_normal((10000, 2))])
    y = 0.05 + np.exp(-8 * np.abs(x[:, 1])) * (1 + np.sin(6 * x[:, 0]) ** 2)

    def geos():
        return [geometry.CubeGeometry("domain", True, [-0.2, -0.5], [1.2, 0.5]),
                geometry.GeometryCoordinates2D("airfoil", False, poly, refine=True, min_refinement_level=8)]

    kw = dict(uniform_level=4, n_cells=6000)
    t_hip = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos(), **kw)
    t_hip.refine()
    monkeypatch.setattr(s_cube, "_make_backend", lambda v, t, k: OracleTreeBackend(v, t, k))
    t_orc = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos(), **kw)
    t_orc.refine()
    assert pt.equal(t_hip.all_centers, t_orc.all_centers) and pt.equal(t_hip.all_levels, t_orc.all_levels)
    assert pt.equal(t_hip.face_ids, t_orc.face_ids) and pt.equal(t_hip.all_nodes, t_orc.all_nodes)
    assert np.array_equal(t_hip._cell_values()["gain"], t_orc._cell_values()["gain"])
    assert int(t_hip.all_levels.max()) >= 8 and len(t_hip.all_levels) > 6000      # geometry refinement overshoots the cap
    # no generated cell lies completely inside the airfoil
    inside = np.array([geos()[1].check_cell(t_hip.all_nodes[f.long()]) for f in t_hip.face_ids[:2000]])
    assert not inside.any()


def test_pre_select_quirk_gpu():
    """reference quirk (s_cube.py:1832-1836): with pre_select=True no cell is ever removed"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    x = cloud(5, 3000, [0, 0], [1, 1])
    y = wake_metric(x, [0.3, 0.5])
    geos = [geometry.CubeGeometry("domain", True, [0, 0], [1, 1]), geometry.SphereGeometry("hole", False, [0.5, 0.5], 0.3)]
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, uniform_level=3, pre_select=True)
    tree._refine_uniform()
    assert len(tree._leaf_cells) == 64 and (tree._topo.first_child != -2).all()


def test_bench_small_workload_matches_reference():
    """the reduced bench workload (299 502 points, 3-D, cylinder): leaf centres bit-exact against the real reference
    (tools/reference_timing.py, run in the dev container: 12.3 s there for refine())"""
    import bench
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    cfg = dict(bench.WORKLOADS["cylinder3D_small"])
    x, metric = bench.synthetic_cylinder3d(cfg)
    geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
            geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"],
                               min_metric=cfg["min_metric"])
    tree.refine()
    z = load("bench_small_centers")
    assert np.array_equal(tree.all_centers.numpy(), z["all_centers"])


@pytest.mark.parametrize("t,ncomp,on_gpu,n,nc", [(8, 1, False, 20000, 3000), (12, 3, False, 20000, 3000), (8, 2, True, 20000, 3000),
                                                 (5, 1, True, 20000, 3000), (8, 1, False, 150000, 400), (5, 3, False, 150000, 400),
                                                 (8, 2, True, 150000, 400)])
def test_export_fit_paths_gpu(tmp_path, t, ncomp, on_gpu, n, nc):
    """ExportData._fit_data through every transport: staged upload into padded rows + LDS-tiled kernel (rows 16-byte
    aligned), direct kernel (ragged rows), data already on the GPU, interpolation at the vertices, batches, and the
    upload of the referenced source rows only when the grid is sparse -- against the oracle"""
    import types
    from sparsespatialsampling_amd.export import ExportData
    from oracle import s3_oracle as orc
    rng = np.random.default_rng(t * 10 + ncomp)
    nv = nc // 2                                                 # n = 150000: the grid references a fraction of the points
    x = rng.random((n, 3))
    centers, vertices = rng.random((nc, 3)), rng.random((nv, 3))
    s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=pt.from_numpy(vertices),
                              levels=None, metric=pt.from_numpy(rng.random(n)), size_initial_cell=1.0,
                              save_path=str(tmp_path), save_name="c", grid_name="g")
    ex = ExportData(s, write_times=[str(i) for i in range(2 * t)], interpolate_at_vertices=True)
    idx_c, dist_c = orc.knn(x, centers, 26)
    idx_v, dist_v = orc.knn(x, vertices, 26)
    w_c, w_v = orc.idw_weights(dist_c), orc.idw_weights(dist_v)
    for b in range(2):                                       # two batches reuse the cached tables / plans
        data = rng.standard_normal((n, ncomp, t)).astype(np.float32)
        d = pt.from_numpy(data).cuda() if on_gpu else pt.from_numpy(data)
        ex._fit_data(pt.from_numpy(x), d, "f", 2 * t)
        for got, ref in ((ex._interpolated_fields.centers, orc.interp(w_c, idx_c, data)),
                         (ex._interpolated_fields.vertices, orc.interp(w_v, idx_v, data))):
            assert tuple(got.shape) == ref.shape and not got.is_cuda
            assert np.abs(got.numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
            assert got[:, :, 0].is_contiguous()              # snapshot-major memory: what the writer stores per time
    assert ex._snapshot_counter == 2 * t
    assert (ex._used_rows is not None) == (n > 100000)          # sparse grid: only the referenced source rows go up
    if ex._used_rows is not None:
        assert ex._used_rows.numel() == len(np.unique(np.concatenate([idx_c.ravel(), idx_v.ravel()])))


@pytest.mark.parametrize("t,ncomp,on_gpu,n,chunk", [(200, 1, False, 120000, 100), (130, 3, False, 120000, 60), (96, 1, True, 120000, 100),
                                                    (257, 1, False, 20000, 100), (64, 2, False, 120000, 30), (101, 1, True, 20000, 100),
                                                    # batch lengths whose last piece used to be 1 .. 3 snapshots long: a device
                                                    # batch, k = 26, sparse grid -- the tail joins the piece before it
                                                    (97, 1, True, 120000, 100), (129, 1, True, 120000, 100), (99, 1, True, 120000, 100)])
def test_export_pipeline_pieces_equal_the_single_piece_gpu(tmp_path, t, ncomp, on_gpu, n, chunk, monkeypatch):
    _pipeline_pieces_case(tmp_path, t, ncomp, on_gpu, n, chunk, monkeypatch)


def test_export_pipeline_pieces_through_the_gathered_copy_gpu(tmp_path, monkeypatch):
    """the same with device-resident pieces gathered into the pitched copy first (S3_EXPORT_INPLACE=0: the path of plans that
    cannot read a table where it lies)"""
    monkeypatch.setenv("S3_EXPORT_INPLACE", "0")
    _pipeline_pieces_case(tmp_path, 96, 1, True, 120000, 100, monkeypatch)


def _pipeline_pieces_case(tmp_path, t, ncomp, on_gpu, n, chunk, monkeypatch):
    """one ``export()`` call pipelined over pieces of the snapshot axis (upload of piece j + 1 / kernel of piece j / download
    of piece j - 1 on three streams, ``chunk_size`` sets the piece length; reference export.py:128-167, 463-467) gives the
    bits of the one-piece sequence, for host and device-resident batches, scalar and vector fields, centres and vertices,
    dense and sparse grids, piece lengths that do not divide the batch"""
    import types
    from sparsespatialsampling_amd.export import ExportData
    rng = np.random.default_rng(t + ncomp)
    nc, nv = 3000, 1500
    x = rng.random((n, 3))
    centers, vertices = rng.random((nc, 3)) * 0.3, rng.random((nv, 3)) * 0.3     # n = 120000: a sparse grid (referenced rows only)
    results = {}
    for mode in ("1", "0"):
        # (host batches are cut into pieces on request only: S3_EXPORT_PIPELINE=host; device-resident ones by default)
        monkeypatch.setenv("S3_EXPORT_PIPELINE", {"1": "1" if on_gpu else "host", "0": "0"}[mode])
        s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=pt.from_numpy(vertices),
                                  levels=None, metric=pt.from_numpy(np.ones(n)), size_initial_cell=1.0,
                                  save_path=str(tmp_path), save_name="c", grid_name="g")
        ex = ExportData(s, write_times=[str(i) for i in range(2 * t)], interpolate_at_vertices=True)
        ex._chunk_size = chunk
        got = []
        data_rng = np.random.default_rng(7)
        for b in range(2):
            data = pt.from_numpy(data_rng.standard_normal((n, ncomp, t)).astype(np.float32))
            ex._fit_data(pt.from_numpy(x), data.cuda() if on_gpu else data, "f", 2 * t)
            got.append((ex._interpolated_fields.centers.clone(), ex._interpolated_fields.vertices.clone()))
            assert ex._interpolated_fields.centers[:, :, 0].is_contiguous()
        pieces = ex._snapshot_pieces(data.cuda() if on_gpu else data, ncomp, t)
        assert (len(pieces) > 1) == (mode == "1" and not (on_gpu and ncomp != 1)), pieces
        assert pieces[0][0] == 0 and pieces[-1][1] == t and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        results[mode] = got
    for (c1, v1), (c0, v0) in zip(results["1"], results["0"]):
        assert pt.equal(c1, c0) and pt.equal(v1, v0)


def test_c1_cylinder2d_full_size_matches_reference():
    """BASELINE config C1 at full size (14 350 points, 87 adaptive iterations, body refined to level 9): grid, faces and
    iteration histories against the real reference (18 s there, tests/golden/gen_golden.py c1)"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    from inputs import c1_cylinder2d
    z = load("c1_cylinder2d")
    x, m, geos, kw = c1_cylinder2d(geometry)
    assert sha(x, m) == str(z["input_sha"])
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(m), geometry_obj=geos, **kw)
    tree.refine()
    assert np.array_equal(tree.all_centers.numpy(), z["all_centers"])
    assert np.array_equal(tree.all_levels.numpy(), z["all_levels"].astype(np.int64))
    assert np.array_equal(tree.face_ids.numpy(), z["face_ids"])
    assert np.array_equal(np.array(tree._n_cells_log), z["n_cells_log"])
    np.testing.assert_allclose(np.array(tree._metric), z["metric_hist"], rtol=1e-12)
    assert tree.data_final_mesh["iterations"] == int(z["iterations"])


def test_c2_oat15_full_size_matches_reference():
    """BASELINE config C2 at full size (3*10^5 clustered points, refined NACA outline as GeometryCoordinates2D,
    ``n_cells_max`` stopping; metric = std_t(p) + std_t(|U|) over 2000 snapshots as the reference's OAT15 script computes it):
    grid of the real reference (tests/golden/gen_golden.py c2; 28 232 cells) as checksums, bit-exact.  The outline predicate
    on the reference side is the generator's shapely stand-in (ref_stubs._Polygon), so GEOS semantics are not what this
    pins -- the adversarial predicate cases are in tests/test_polygon_predicate.py.  Then the interpolation at the
    config's size: 2000 snapshots of a scalar and of a 2-component field through the tiled kernel."""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry, hipops, metrics
    from inputs import c2_fields, c2_oat15
    from oracle import s3_oracle as orc
    z = load("c2_oat15")
    x, m, geos, kw = c2_oat15(geometry, z["metric_f16"])
    assert sha(x, m) == str(z["input_sha"])
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(m), geometry_obj=geos, **kw)
    tree.refine()
    centers, levels = tree.all_centers.numpy(), tree.all_levels.numpy()
    assert len(centers) == int(z["n_leaf"]) == 28232
    assert np.array_equal(centers[:64], z["head_centers"]) and np.array_equal(tree.face_ids.numpy()[:64], z["head_faces"])
    assert np.array_equal(np.bincount(levels.reshape(-1)), z["level_hist"])
    assert sha(centers) == str(z["sha_centers"]) and sha(levels.astype(np.int64)) == str(z["sha_levels"])
    assert sha(tree.face_ids.numpy().astype(np.int32)) == str(z["sha_faces"]) and sha(tree.all_nodes.numpy()) == str(z["sha_nodes"])
    assert np.array_equal(np.array(tree._n_cells_log), z["n_cells_log"])
    np.testing.assert_allclose(np.array(tree._metric), z["metric_hist"], rtol=1e-12)
    assert tree.data_final_mesh["iterations"] == int(z["iterations"])
    tree.close()

    # the metric upstream (metrics.temporal_std on the device) on the same synthetic fields, 200 of the 2000 snapshots:
    # float64 torch on the host is the reference's own computation
    p, u = c2_fields(x[:20000], 0, 200)
    got = metrics.temporal_std(pt.from_numpy(p).cuda()).reshape(-1) + metrics.temporal_std(pt.from_numpy(u).cuda().norm(dim=1, keepdim=True)).reshape(-1)
    want = pt.from_numpy(p)[:, 0].double().std(-1) + pt.from_numpy(u).double().norm(dim=1).std(-1)
    assert pt.allclose(got.cpu(), want, rtol=1e-5, atol=1e-7)       # |U| is formed in float32 on the device

    # interpolation at the config's size (k = 8): scalar p [N, 1, 2000] and vector U [N, 2, 2000]
    k, n, nc = 8, len(x), len(centers)
    knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 2))
    idx, dist = knn.query(centers, k)
    knn.close()
    w = hipops.idw_weights(dist)
    plan = hipops.InterpPlan(idx, n, centers)
    sel = np.random.default_rng(0).choice(nc, 300, replace=False)
    for ncomp in (1, 2):
        row_len = ncomp * 2000
        data = hipops.padded_rows(n, row_len, pt.float32, "cuda")
        data.normal_()
        out = plan.interp(w, data)
        assert pt.equal(out, hipops.interp(w, idx, data.contiguous()))
        const = hipops.padded_rows(n, row_len, pt.float32, "cuda")
        const.fill_(0.75)
        assert pt.allclose(plan.interp(w, const), pt.full((nc, row_len), 0.75, dtype=pt.float64, device="cuda"), rtol=1e-13, atol=0)
        # oracle on a slice of the cells (the rows they reference only)
        i_s, w_s = idx[sel].cpu().numpy(), w[sel].cpu().numpy()
        rows, inv = np.unique(i_s, return_inverse=True)
        sub = data[pt.from_numpy(rows).cuda().long()].cpu().numpy().reshape(len(rows), ncomp, 2000)
        ref = orc.interp(w_s, inv.reshape(i_s.shape), sub).reshape(len(sel), row_len)
        assert np.abs(out[sel].cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()


def _oracle_grid(x, metric, geos, **kw):
    """the same ``SamplingTree.refine()`` with the CPU oracle's kernels behind it (tests/oracle_backend.py, bucket-grid queries) ->
    (SHA-256 over centres / levels / faces / nodes, metric history): the full-size checker where the reference itself cannot finish
    (C2 already takes it 8 min 45 s)"""
    import bench
    import sparsespatialsampling_amd.s_cube as s_cube
    from tests.oracle_backend import OracleTreeBackend
    product = s_cube._make_backend
    s_cube._make_backend = lambda v, t, k: OracleTreeBackend(v, t, k, grid=True)
    try:
        tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
        tree.refine()
        return bench.grid_sha(tree.all_centers, tree.all_levels, tree.face_ids, tree.all_nodes), np.array(tree._metric), len(tree.all_centers)
    finally:
        s_cube._make_backend = product


def test_c4_sub_box_grid_equals_the_cpu_port_cell_for_cell():
    """C4's cloud restricted to one octant of the unit box (the points with every coordinate below 0.5: 6.25 * 10^6 of the 5 * 10^7,
    cell budget 1.25 * 10^6 = an eighth of C4's): the grid of the HIP backend and the grid of the CPU port (oracle kernels, same
    host logic) are the same cell for cell -- centres, levels, face ids and node coordinates hash to the same SHA-256 (VERDICT r4:
    the full-size C3 / C4 grids were only checked through properties)"""
    import bench
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    cfg = dict(bench.WORKLOADS["box5e7"])
    x, metric = bench.synthetic_box(cfg)
    keep = (x < 0.5).all(1)
    x, metric = np.ascontiguousarray(x[keep]), np.ascontiguousarray(metric[keep])
    assert 6_200_000 < len(x) < 6_300_000
    geos = [geometry.CubeGeometry("domain", True, [0.0, 0.0, 0.0], [0.5, 0.5, 0.5])]
    kw = dict(uniform_level=cfg["uniform_levels"], n_cells=cfg["n_cells_max"] // 8)
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
    tree.refine()
    sha_gpu = bench.grid_sha(tree.all_centers, tree.all_levels, tree.face_ids, tree.all_nodes)
    n_gpu, hist = len(tree.all_centers), np.array(tree._metric)
    assert n_gpu >= cfg["n_cells_max"] // 8 and tree.face_ids.dtype == pt.int32
    tree.close()
    sha_cpu, hist_cpu, n_cpu = _oracle_grid(x, metric, geos, **kw)
    assert n_cpu == n_gpu and sha_cpu == sha_gpu
    np.testing.assert_allclose(hist, hist_cpu, rtol=1e-12)


def test_c4_box5e7_full_size_properties():
    """BASELINE config C4 at full size on one GPU (5*10^7 random centroids in the unit box, ``n_cells_max`` = 10^7, three
    scalar fields in batches of 16 snapshots): grid properties, stopping rule, planned == direct, constant reproduction,
    linearity, and the oracle on a slice of the cells -- for 64-byte rows (one scalar field), for the three fields packed
    as one 3-component batch (192-byte rows) and for a ragged 25-snapshot batch."""
    import bench
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry, hipops
    from oracle import s3_oracle as orc
    cfg = dict(bench.WORKLOADS["box5e7"])
    x, metric = bench.synthetic_box(cfg)
    geos = [geometry.CubeGeometry("domain", True, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])]
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"],
                               n_cells=cfg["n_cells_max"])
    del metric
    tree.refine()
    centers, nodes = tree.all_centers.numpy(), tree.all_nodes.numpy()
    faces, levels = tree.face_ids.numpy(), tree.all_levels.numpy().reshape(-1)
    nc = len(centers)
    assert nc == 10_062_144 and faces.shape == (nc, 8) and tree.face_ids.dtype == pt.int32
    assert tree._n_cells_log[-1] >= cfg["n_cells_max"] > tree._n_cells_log[-2]              # stopped by the cell budget
    tree.close()
    blk = slice(0, nc, 7)                                                                 # every 7th leaf: 1.4*10^6 cells
    corner = nodes[faces[blk].astype(np.int64)]
    assert np.abs(corner.mean(1) - centers[blk]).max() <= 4e-16
    assert np.allclose(corner.max(1) - corner.min(1), (1.0 / 2.0 ** levels[blk])[:, None], rtol=1e-12, atol=0)
    key = np.round(centers * 2.0 ** 12).astype(np.int64)                                  # centres are dyadic: exact keys
    key = (key[:, 0] << 28) | (key[:, 1] << 14) | key[:, 2]
    assert len(np.unique(key)) == nc
    assert centers.min() > 0 and centers.max() < 1 and levels.min() >= cfg["uniform_levels"]
    del corner, key, nodes, faces

    k, n = 26, len(x)
    knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
    idx, dist = knn.query(centers, k)
    knn.close()
    w = hipops.idw_weights(dist)
    del dist
    assert pt.allclose(w.sum(1), pt.ones(nc, dtype=pt.float64, device="cuda"), rtol=1e-14, atol=0)
    plan = hipops.InterpPlan(idx, n, centers)
    sel = np.random.default_rng(1).choice(nc, 10_000, replace=False)
    i_s, w_s = idx[sel].cpu().numpy(), w[sel].cpu().numpy()
    rows, inv = np.unique(i_s, return_inverse=True)
    rows_dev = pt.from_numpy(rows).cuda().long()
    for ncomp, t in ((1, 16), (3, 16), (1, 25)):
        row_len = ncomp * t
        a, b = hipops.padded_rows(n, row_len, pt.float32, "cuda"), hipops.padded_rows(n, row_len, pt.float32, "cuda")
        a.normal_(), b.normal_()
        fa = plan.interp(w, a)
        assert pt.equal(fa, hipops.interp(w, idx, a.contiguous()))                        # planned == direct, bit for bit
        fb = plan.interp(w, b)
        both = hipops.padded_rows(n, row_len, pt.float64, "cuda")
        both.copy_(a.double() + 2 * b.double())
        assert pt.allclose(plan.interp(w, both), fa + 2 * fb, rtol=1e-12, atol=1e-12)
        del both, fb
        b.fill_(3.25)
        assert pt.allclose(plan.interp(w, b), pt.full((nc, row_len), 3.25, dtype=pt.float64, device="cuda"), rtol=1e-13, atol=0)
        sub = a[rows_dev].cpu().numpy().reshape(len(rows), ncomp, t)
        ref = orc.interp(w_s, inv.reshape(i_s.shape), sub).reshape(len(sel), row_len)
        assert np.abs(fa[sel].cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
        del a, b, fa


def test_full_size_c3_properties():
    """BASELINE's bench configuration at full size (4 991 774 points -> 461 130 cells, k = 26), checked through
    size-independent properties: grid geometry (centres = mean of the cell's vertices, edge length = width / 2^level,
    unique leaves, body cells removed, surface resolved), stopping rule, and -- for the planned interpolation on the real
    neighbour table -- reproduction of a constant field, linearity in the data and agreement with the direct kernel"""
    import bench
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry, hipops
    cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
    x, metric = bench.synthetic_cylinder3d(cfg)
    geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
            geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"],
                               min_metric=cfg["min_metric"])
    tree.refine()
    centers, nodes = tree.all_centers.numpy(), tree.all_nodes.numpy()
    faces, levels = tree.face_ids.numpy().astype(np.int64), tree.all_levels.numpy().reshape(-1)
    nc = len(centers)
    assert nc == 461_130 and faces.shape == (nc, 8) and tree.face_ids.dtype == pt.int32
    corner = nodes[faces]                                                   # [nc, 8, 3]
    assert np.abs(corner.mean(1) - centers).max() <= 1e-15 * 4
    edge = corner.max(1) - corner.min(1)
    assert np.allclose(edge, (float(tree._width) / 2.0 ** levels)[:, None], rtol=1e-12, atol=0)
    assert len(np.unique(centers, axis=0)) == nc
    # no leaf lies completely inside the cylinder, and the cells cut by its surface sit on the finest level found there
    r = np.hypot(corner[..., 0] - 0.8, corner[..., 1] - 1.0)
    inside = r <= 0.05
    assert not inside.all(1).any()
    cut = inside.any(1) & ~inside.all(1)
    assert cut.any() and len(np.unique(levels[cut])) == 1 and levels[cut][0] == levels.max()
    assert tree._metric[-1] >= cfg["min_metric"] > tree._metric[-2]        # stopped by the metric, not earlier
    # ... and cell for cell the grid of the CPU port (oracle kernels behind the same host logic; ~3 s on the box's host cores):
    # centres, levels, face ids and node coordinates hash to the same SHA-256 (what bench.py prints as same_grid_sha)
    sha_cpu, hist_cpu, n_cpu = _oracle_grid(x, metric, geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
    assert n_cpu == nc and sha_cpu == bench.grid_sha(tree.all_centers, tree.all_levels, tree.face_ids, tree.all_nodes)
    np.testing.assert_allclose(np.array(tree._metric), hist_cpu, rtol=1e-12)

    k, t = 26, 64
    knn = hipops.KnnIndex(x)
    idx, dist = knn.query(centers, k)
    knn.close()
    w = hipops.idw_weights(dist)
    assert pt.allclose(w.sum(1), pt.ones(nc, dtype=pt.float64, device="cuda"), rtol=1e-14, atol=0)
    plan = hipops.InterpPlan(idx, len(x), centers)
    a, b = hipops.padded_rows(len(x), t, pt.float32, "cuda"), hipops.padded_rows(len(x), t, pt.float32, "cuda")
    a.normal_(), b.normal_()
    const = hipops.padded_rows(len(x), t, pt.float32, "cuda")
    const.fill_(-2.5)
    assert pt.allclose(plan.interp(w, const), pt.full((nc, t), -2.5, dtype=pt.float64, device="cuda"), rtol=1e-13, atol=0)
    fa, fb = plan.interp(w, a), plan.interp(w, b)
    both = hipops.padded_rows(len(x), t, pt.float64, "cuda")
    both.copy_(a.double() + 2 * b.double())
    assert pt.allclose(plan.interp(w, both), fa + 2 * fb, rtol=1e-12, atol=1e-12)
    assert pt.equal(fa, hipops.interp(w, idx, a.contiguous()).reshape(nc, t))        # same arithmetic as the direct kernel
    plan.close()
    del a, b, const, both, fa, fb

    # the layout bench.py times: only the referenced rows resident, in Hilbert order, pitch of padded_rows,
    # 1000 snapshots -- against the oracle on a slice of 10^4 cells, and the batch lengths of roofline_batches (25 scalar
    # snapshots: 100-byte ragged rows through the persistent kernel; 3 x 25: 300-byte rows)
    from oracle import s3_oracle as orc
    used, remap = hipops.referenced_rows([idx], len(x), coords=x)
    n_rows = int(used.numel())
    assert n_rows == 2_430_607
    idx_c = idx.clone()
    hipops.remap_indices(idx_c, remap)
    plan = hipops.InterpPlan(idx_c, n_rows, centers)
    plan.set_weights(w)
    rng = np.random.default_rng(8)
    sel = np.sort(rng.choice(nc, 10_000, replace=False))
    i_sel, w_sel = idx_c[pt.from_numpy(sel).cuda()].cpu().numpy(), w[pt.from_numpy(sel).cuda()].cpu().numpy()
    rows_sel, inv = np.unique(i_sel, return_inverse=True)
    for row_len in (1000, 25, 75):
        data = hipops.padded_rows(n_rows, row_len, pt.float32, "cuda")
        data.normal_(generator=pt.Generator(device="cuda").manual_seed(row_len))
        assert data.stride(0) * 4 % 128 == 0 and (row_len != 1000 or data.stride(0) == 35 * 32)
        got = plan.interp(w, data)
        sub = data[pt.from_numpy(rows_sel).cuda().long()].contiguous().cpu().numpy().reshape(len(rows_sel), 1, row_len)
        ref = orc.interp(w_sel, inv.reshape(i_sel.shape), sub).reshape(len(sel), row_len)
        out = got[pt.from_numpy(sel).cuda()].cpu().numpy()
        assert np.abs(out - ref).max() <= 1e-13 * np.abs(ref).max()
        if row_len != 1000:                       # the whole batch against the direct gather kernel, bit for bit
            assert pt.equal(got, hipops.interp(w, idx_c, data.contiguous()))
        del data, got

    # the layout bench.py times since round 4: the dense batch [N_points, T] of ALL points as interpolate_data
    # receives it (export.py:446-468), read in place through the plan's source ids -- the WHOLE 1000-snapshot batch against the
    # direct gather kernel on the same table (another kernel: no tiles, no LDS staging, no phase shifts) bit for bit, the
    # oracle on the slice, and the reference's 25-snapshot batches the same way
    plan.set_source_ids(used.contiguous(), len(x))
    idx_full = used.long()[idx_c.long()].to(pt.int32).contiguous()
    for row_len in (1000, 25):
        table = pt.empty((len(x), row_len), dtype=pt.float32, device="cuda").normal_(generator=pt.Generator(device="cuda").manual_seed(7 + row_len))
        got = plan.interp_src(table)
        assert pt.equal(got, hipops.interp(w, idx_full, table))
        sub = table[used.long()[pt.from_numpy(rows_sel).cuda().long()]].contiguous().cpu().numpy().reshape(len(rows_sel), 1, row_len)
        ref = orc.interp(w_sel, inv.reshape(i_sel.shape), sub).reshape(len(sel), row_len)
        out = got[pt.from_numpy(sel).cuda()].cpu().numpy()
        assert np.abs(out - ref).max() <= 1e-13 * np.abs(ref).max()
        del table, got
    plan.close()


@pytest.mark.parametrize("seed", [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11])
def test_refine_random_configurations_gpu(seed):
    """randomly drawn configurations (dimension, stopping rule, ramp, 2:1 balance, pre_select, bodies of every kind): the
    HIP path against the real reference's grid"""
    from tests.test_tree_host_logic import check_random_case
    check_random_case(seed)


@pytest.mark.parametrize("name", ["refine_2d_metric", "refine_3d_metric"])
def test_geometry_without_kernel_spec_takes_the_host_path_gpu(name):
    """a user-defined geometry (only ``check_cell``, reference s_cube.py:1816-1837) on the HIP backend"""
    from tests.test_tree_host_logic import check_geometry_fallback
    check_geometry_fallback(name)
