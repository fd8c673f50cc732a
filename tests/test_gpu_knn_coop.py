"""The wavefront form of the refine's child-metric search (csrc/knn.hip: child_metric_coop_kernel, then child_metric_near_kernel,
child_metric_far_kernel and the per-lane search for what each leaves over) against the per-lane kernel: same bits.  GPU only."""
import numpy as np
import pytest
import torch as pt

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,k", [(3, 26), (2, 8), (3, 5), (3, 50)])
@pytest.mark.parametrize("cloud", ["uniform", "lattice", "graded"])
def test_wavefront_per_cell_equals_per_lane(monkeypatch, dim, k, cloud):
    """s3_child_gain_reuse on batches of new cells of several levels (fine cells whose child points share one box, coarse ones
    that do not, cells in a hole of the cloud, cells outside it) with S3_KNN_COOP=1 (every cell first to a wavefront) and =0
    (the per-lane search alone): metric of every child point and gain of every cell bit for bit; uniform random cloud,
    a lattice (ties in distance at every turn) and a graded cloud (refined buckets: two-level index)"""
    from sparsespatialsampling_amd import hipops
    rng = np.random.default_rng(dim * 100 + k)
    n_pts = 200_000 if dim == 3 else 60_000
    if cloud == "uniform":
        x = rng.random((n_pts, dim))
        x = x[np.linalg.norm(x - 0.5, axis=1) > 0.12]                    # a hole: searches much wider than a box
    elif cloud == "lattice":
        m = int(round(n_pts ** (1 / dim)))
        x = np.stack(np.meshgrid(*[np.arange(m) / m] * dim, indexing="ij"), -1).reshape(-1, dim)
    else:
        x = np.concatenate([rng.random((n_pts // 2, dim)), 0.5 + 0.01 * rng.standard_normal((n_pts // 2, dim))])
    y = np.sin(7 * x[:, 0]) + x[:, 1] ** 2
    knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, dim))
    knn.set_values(y)
    nch, n, width = 2 ** dim, 3000, 1.0
    lf = hipops.to_device(np.array([1 / nch * ((width / 2 ** lv) ** dim) for lv in range(64)]))
    for lv in (3, 5, 6, 7, 9):
        cap = n + nch
        center = pt.from_numpy(rng.random((cap, dim)) * 1.2 - 0.1).cuda()
        level = pt.full((cap,), lv, dtype=pt.int32, device="cuda")
        parents = pt.from_numpy(rng.integers(0, nch, (n + nch - 1) // nch).astype(np.int32)).cuda()
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("S3_KNN_COOP", mode)
            metric, gain = pt.zeros(cap, dtype=pt.float64, device="cuda"), pt.zeros(cap, dtype=pt.float64, device="cuda")
            child = pt.from_numpy(rng.random((cap, nch))).cuda() if mode == "1" else res["1"][3].clone()
            start = child.clone()
            scratch = pt.zeros(n * (nch + 1) + 2 + n * nch, dtype=pt.float64, device="cuda")
            hipops.child_gain_reuse(knn, k, center, level, nch, n, width, lf, 0.37, metric, gain, scratch, parents, 0, child)
            res[mode] = (child[nch:].clone(), metric[nch:].clone(), gain[nch:].clone(), start)
        assert pt.equal(res["1"][0], res["0"][0]) and pt.equal(res["1"][1], res["0"][1]) and pt.equal(res["1"][2], res["0"][2])
        assert bool(pt.isfinite(res["1"][0]).all())
        # the centre's value is the parent's entry
        want = res["1"][3][parents.long()[pt.arange(n, device="cuda") // nch], pt.arange(n, device="cuda") % nch]
        assert pt.equal(res["1"][1], want)
    knn.close()


@pytest.mark.parametrize("name", ["refine_2d_metric", "refine_2d_delta", "refine_3d_metric", "refine_3d_ncells_cone", "refine_3d_polytopes"])
def test_refine_goldens_through_the_wavefront_kernels(monkeypatch, name):
    """the reference's grids (cell ids / levels / centres / faces / vertices / per-cell metric + gain, bit for bit) with every batch
    forced through the wavefront-per-cell kernel, the streaming search and the per-lane search for what they leave
    (S3_KNN_COOP=1: whatever S3_KNN_COOP_MIN says)"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    from inputs import refine_inputs, sha
    from tests.test_gpu_refine import check_outputs_against_golden, check_tree_against_golden, load
    monkeypatch.setenv("S3_KNN_COOP", "1")
    z = load(name)
    x, y, geos, kw = refine_inputs(name, geometry)
    assert sha(x, y) == str(z["input_sha"])
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
    tree.refine()
    check_tree_against_golden(tree, z)
    check_outputs_against_golden(tree, z)


def test_c1_full_size_through_the_wavefront_kernels(monkeypatch):
    """BASELINE config C1 at full size (a structured-looking 2-D cloud, 87 adaptive iterations, body refined to level 9) with
    S3_KNN_COOP=1: the reference's grid"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    from inputs import c1_cylinder2d, sha
    from tests.test_gpu_refine import load
    monkeypatch.setenv("S3_KNN_COOP", "1")
    z = load("c1_cylinder2d")
    x, m, geos, kw = c1_cylinder2d(geometry)
    assert sha(x, m) == str(z["input_sha"])
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(m), geometry_obj=geos, **kw)
    tree.refine()
    assert np.array_equal(tree.all_centers.numpy(), z["all_centers"])
    assert np.array_equal(tree.all_levels.numpy(), z["all_levels"].astype(np.int64))
    assert np.array_equal(tree.face_ids.numpy(), z["face_ids"])
    np.testing.assert_allclose(np.array(tree._metric), z["metric_hist"], rtol=1e-12)
