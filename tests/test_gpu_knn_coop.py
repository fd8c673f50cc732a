"""The wavefront-per-cell form of the refine's child-metric search (csrc/knn.hip: child_metric_coop_kernel + the per-lane
search for what it leaves over) against the per-lane kernel: same bits.  GPU only."""
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
            scratch = pt.zeros(n * (nch + 1) + 2 + (n * nch + 1) // 2, dtype=pt.float64, device="cuda")
            hipops.child_gain_reuse(knn, k, center, level, nch, n, width, lf, 0.37, metric, gain, scratch, parents, 0, child)
            res[mode] = (child[nch:].clone(), metric[nch:].clone(), gain[nch:].clone(), start)
        assert pt.equal(res["1"][0], res["0"][0]) and pt.equal(res["1"][1], res["0"][1]) and pt.equal(res["1"][2], res["0"][2])
        assert bool(pt.isfinite(res["1"][0]).all())
        # the centre's value is the parent's entry
        want = res["1"][3][parents.long()[pt.arange(n, device="cuda") // nch], pt.arange(n, device="cuda") % nch]
        assert pt.equal(res["1"][1], want)
    knn.close()
