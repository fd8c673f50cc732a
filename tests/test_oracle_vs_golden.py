"""
Pins the CPU oracle (oracle/s3_oracle.c) against golden vectors produced by the REAL reference
(tests/golden/gen_golden.py).  CPU only.
"""
import ctypes
import os
import types

import numpy as np
import pytest
import torch as pt

from oracle import s3_oracle as orc
from inputs import POLYTOPES, mask_cells, polytope, polytope_cells, refine_inputs, sha

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# geometry "module" that just records constructor arguments (the oracle tests need inputs, not predicates)
SPEC = types.SimpleNamespace(
    CubeGeometry=lambda *a, **k: ("cube", a, k),
    SphereGeometry=lambda *a, **k: ("sphere", a, k),
    CylinderGeometry3D=lambda *a, **k: ("cylinder", a, k),
)


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


@pytest.mark.parametrize("name", ["interp_k8_c1_f32", "interp_k8_c3_f64", "interp_k26_c1_f32", "interp_k26_c3_f32",
                                  "interp_k26_c1_f64"])
def test_interp(name):
    z = load(name)
    out = orc.interp(z["w"], z["idx"], z["data"])
    assert out.shape == z["out"].shape and out.dtype == np.float64
    scale = np.abs(z["out"]).max()
    assert np.abs(out - z["out"]).max() <= 1e-14 * scale      # reference export.py:467; contract in BASELINE: 1e-5


@pytest.mark.parametrize("name,k", [("knncache_2d", 8), ("knncache_3d", 26)])
def test_knn_cache(name, k):
    z = load(name)
    for q, idx_ref, w_ref in ((z["centers"], z["idx_c"], z["w_c"]), (z["vertices"], z["idx_v"], z["w_v"])):
        idx, dist = orc.knn(z["coords"], q, k)
        assert np.array_equal(idx, idx_ref)                     # bit-exact neighbour indices (export.py:425)
        w = orc.idw_weights(dist)
        assert np.array_equal(w, w_ref)                         # bit-exact weights incl. clamp rows (export.py:428-429)
    idx, dist = orc.knn(z["coords"], z["centers"], k)
    assert np.array_equal(dist, z["dist_c"])
    assert dist[5, 0] == 0.0 and dist[6, 0] == 0.0              # the exact-hit rows are really exercised


@pytest.mark.parametrize("name,k", [("predict_2d", 8), ("predict_3d", 26)])
def test_predict(name, k):
    z = load(name)
    pred = orc.idw_predict(z["x"], z["y"], z["q"], k)
    assert np.array_equal(pred, z["pred"])                      # bit-exact incl. the indicator-weight rows
    assert pred[3] == z["y"][10] and pred[4] == z["y"][11]


def test_masks():
    z = load("masks")
    rng = np.random.default_rng(7)
    c2, h2 = mask_cells(2, 400, rng)
    c3, h3 = mask_cells(3, 400, rng)
    assert np.array_equal(c2[:, None, :] + orc.DIRS[2][None] * h2[:, None, None], z["cells2"])
    assert np.array_equal(c3[:, None, :] + orc.DIRS[3][None] * h3[:, None, None], z["cells3"])

    # the oracle API is (centre, level, width) with node offset (0.5*width)/2^level; level 0 and width = 2h give
    # offset == h exactly, so the nodes are bit-identical to the fixture's
    def run(fn, c, h, *args):
        out = np.zeros(len(c), dtype=bool)
        lv = np.zeros(1, dtype=np.int32)
        for i in range(len(c)):
            out[i] = fn(c[i:i + 1], lv, 2.0 * h[i], *args)[0]
        return out

    for ki in (True, False):
        for rm in (False, True):
            sfx = f"_{int(ki)}_r{int(rm)}"
            assert np.array_equal(run(orc.mask_box, c2, h2, [0.0, 0.1], [1.0, 0.9], rm, ki), z["cube2" + sfx])
            assert np.array_equal(run(orc.mask_box, c3, h3, [0.0, 0.1, -0.2], [1.0, 0.9, 0.7], rm, ki),
                                  z["cube3" + sfx])
            assert np.array_equal(run(orc.mask_sphere, c2, h2, [0.4, 0.5], 0.45, rm, ki), z["sphere2" + sfx])
            assert np.array_equal(run(orc.mask_sphere, c3, h3, [0.4, 0.5, 0.3], 0.6, rm, ki), z["sphere3" + sfx])
            assert np.array_equal(run(orc.mask_cylinder, c3, h3, [(0.2, 0.3, -0.1), (0.9, 0.6, 0.8)], 0.35, rm, ki),
                                  z["cyl3" + sfx])
            assert np.array_equal(run(orc.mask_cylinder, c3, h3, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], [0.5, 0.1], rm, ki),
                                  z["cone3" + sfx])
            assert np.array_equal(run(orc.mask_polygon, c2, h2, z["poly"], rm, ki), z["poly2" + sfx])
    assert 0 < z["cyl3_0_r0"].sum() < 400 and 0 < z["poly2_1_r1"].sum() < 400     # non-trivial truth tables


@pytest.mark.parametrize("key", sorted(POLYTOPES))
def test_masks_polytopes(key):
    """triangle / prism / tetrahedron / pyramid: (i) the package's geometry classes give the reference's verdict per cell
    (host check_cell), (ii) the oracle predicate fed with their kernel_spec() does too -- incl. lattice cells whose nodes
    lie exactly on faces, edges and corners of the dyadic bodies"""
    from sparsespatialsampling_amd import geometry
    z = load("masks_polytopes")
    d = 2 if key.startswith("tri") else 3
    c, h = polytope_cells(2, np.random.default_rng(11)) if d == 2 else _second(polytope_cells, 11)
    assert np.array_equal(c, z[f"c{d}"]) and np.array_equal(h, z[f"h{d}"])
    nodes = c[:, None, :] + orc.DIRS[d][None] * h[:, None, None]
    lv = np.zeros(1, dtype=np.int32)
    for ki in (True, False):
        g = polytope(geometry, key, ki)
        spec = g.kernel_spec()
        fn = {"triangle": orc.mask_triangle, "prism": orc.mask_prism, "tetrahedra": orc.mask_tetrahedra}[spec[0]]
        for rm in (False, True):
            want = z[f"{key}_{int(ki)}_r{int(rm)}"]
            host = np.array([g.check_cell(pt.from_numpy(nodes[i]), rm) for i in range(len(nodes))])
            assert np.array_equal(host, want), (key, ki, rm, "host class")
            got = np.array([fn(c[i:i + 1], lv, 2.0 * h[i], *spec[1:], rm, ki)[0] for i in range(len(c))])
            assert np.array_equal(got, want), (key, ki, rm, "oracle")
    assert 0 < z[f"{key}_0_r0"].sum() < len(c) and 0 < z[f"{key}_1_r1"].sum() < len(c)       # non-trivial tables


def _second(fn, seed):
    """the 3-D cell stream of the fixture is drawn after the 2-D one from the same generator"""
    rng = np.random.default_rng(seed)
    fn(2, rng)
    return fn(3, rng)


@pytest.mark.parametrize("name,k", [("refine_2d_metric", 8), ("refine_2d_delta", 8), ("refine_3d_metric", 26),
                                    ("refine_3d_delta", 26), ("refine_3d_ncells_cone", 26)])
def test_child_gain_trace(name, k):
    """metric + gain of every cell the reference created (s_cube.py:207-241,1859) -- bit exact."""
    z = load(name)
    x, y, _, _ = refine_inputs(name, SPEC)
    assert sha(x, y) == str(z["input_sha"])
    width, gain0 = float(z["width"]), float(z["gain0"])
    sel = np.arange(1, len(z["level"]))
    metric, gain = orc.child_gain(x, y, k, z["center"][sel], z["level"][sel], width, gain0)
    assert np.array_equal(metric[:, 0], z["metric"][sel])
    valid = z["state"][sel] != 2                                  # invalid cells get gain = 0 (s_cube.py:723)
    assert np.array_equal(gain[valid], z["gain"][sel][valid])
    assert np.all(z["gain"][sel][~valid] == 0)


@pytest.mark.parametrize("name,k", [("knncache_2d", 8), ("knncache_3d", 26)])
def test_grid_knn_cache(name, k):
    """the bucket-grid form of the query (the oracle's stand-in for the reference's kd-tree, used where brute force cannot run:
    bench.py's CPU baselines at full size) against the same goldens as the brute-force form: bit-exact indices, distances,
    weights, incl. the exact-hit rows and queries outside the cloud's bounding box"""
    z = load(name)
    for occ in (0.5, 3.0, 40.0):
        grid = orc.GridIndex(z["coords"], occ)
        for q, idx_ref, w_ref in ((z["centers"], z["idx_c"], z["w_c"]), (z["vertices"], z["idx_v"], z["w_v"])):
            idx, dist = grid.knn(q, k)
            assert np.array_equal(idx, idx_ref)
            assert np.array_equal(orc.idw_weights(dist), w_ref)
        idx, dist = grid.knn(z["centers"], k)
        assert np.array_equal(dist, z["dist_c"])
        grid.close()


@pytest.mark.parametrize("name,k", [("predict_2d", 8), ("predict_3d", 26)])
def test_grid_predict(name, k):
    z = load(name)
    grid = orc.GridIndex(z["x"])
    assert np.array_equal(grid.idw_predict(z["y"], z["q"], k), z["pred"])
    grid.close()


@pytest.mark.parametrize("name,k", [("refine_2d_metric", 8), ("refine_3d_metric", 26), ("refine_3d_ncells_cone", 26)])
def test_grid_child_gain_trace(name, k):
    z = load(name)
    x, y, _, _ = refine_inputs(name, SPEC)
    width, gain0 = float(z["width"]), float(z["gain0"])
    sel = np.arange(1, len(z["level"]))
    grid = orc.GridIndex(x)
    metric, gain = grid.child_gain(y, k, z["center"][sel], z["level"][sel], width, gain0)
    assert np.array_equal(metric[:, 0], z["metric"][sel])
    valid = z["state"][sel] != 2
    assert np.array_equal(gain[valid], z["gain"][sel][valid])
    grid.close()


def test_grid_equals_brute_force_on_hard_clouds():
    """ties (lattice points), clustered clouds, duplicate points, queries far outside, k = n"""
    rng = np.random.default_rng(5)
    for d in (2, 3):
        lattice = np.stack(np.meshgrid(*[np.arange(9.0)] * d, indexing="ij"), -1).reshape(-1, d)
        clustered = np.concatenate([rng.random((3000, d)), 0.5 + 1e-3 * rng.standard_normal((3000, d)), np.zeros((5, d))])
        for pts in (lattice, clustered, rng.random((7, d))):
            q = np.concatenate([pts[:50] + 0.0, rng.random((200, d)) * 3 - 1, lattice[:40] + 0.5])
            for k in (1, 5, min(26, len(pts)), min(64, len(pts))):
                grid = orc.GridIndex(pts, float(rng.choice([0.3, 2.0, 10.0])))
                i0, d0 = orc.knn(pts, q, k)
                i1, d1 = grid.knn(q, k)
                assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
                grid.close()


def test_child_centers_recurrence():
    """centre(child) = centre(parent) + dir * (0.25*width)/2^level(parent)  (s_cube.py:441) -- bit exact."""
    for name in ("refine_2d_metric", "refine_3d_metric"):
        z = load(name)
        d = z["center"].shape[1]
        width = float(z["width"])
        par = z["parent"][1:]
        child_no = (np.arange(1, len(par) + 1) - 1) % (2 ** d)       # children are created consecutively
        off = (0.25 * width) / (2.0 ** z["level"][par])
        expect = z["center"][par] + orc.DIRS[d][child_no] * off[:, None]
        assert np.array_equal(expect, z["center"][1:])


def test_topn_matches_heapq():
    import heapq
    rng = np.random.default_rng(3)
    g = np.round(rng.random(500), 2)                              # many exact ties -> exercises the -idx tie rule
    ids = rng.permutation(5000)[:500]
    gain_of = dict(zip(ids.tolist(), g.tolist()))
    ref = heapq.nlargest(37, set(ids.tolist()), key=lambda i: (gain_of[i], -i))      # s_cube.py:601-602
    assert orc.topn(g, ids, 37).tolist() == ref


def test_sum_orders():
    import torch
    rng = np.random.default_rng(5)
    for n in (4, 8, 26):
        a = rng.random((64, n))
        t = torch.from_numpy(a).sum(dim=1).numpy()
        mine = np.array([orc.lib().s3o_torch_inner_sum(np.ascontiguousarray(r).ctypes.data_as(ctypes.c_void_p), n)
                         for r in a])
        assert np.array_equal(t, mine)


def test_sklearn_brute_force_branch_is_matched_closely():
    """scikit-learn leaves the kd-tree for brute force when k >= N // 2 (sklearn/neighbors/_base.py:625-633; only toy clouds get
    there, e.g. the reference's 50-point unit tests): squared distances from |q|^2 + |p|^2 - 2 q.p through a BLAS product instead of
    the sum of squared differences.  The exact search of this build (and of the oracle) is NOT that arithmetic; the fixture
    (tests/golden/gen_sklearn_brute.py, generated with scikit-learn itself) pins how far apart they are: same neighbours in the same
    order, distances equal to 1e-13 absolute on unit-sized clouds (the expansion cancels for close pairs: 3e-12 relative at a
    distance of 0.0075), weights="distance" predictions to 1e-10 relative."""
    z = np.load(os.path.join(G, "sklearn_brute.npz"))
    for i in range(int(z["n_cases"])):
        assert str(z[f"method{i}"]) == "brute"
        x, y, q, k = z[f"x{i}"], z[f"y{i}"], z[f"q{i}"], int(z[f"k{i}"])
        idx, dist = orc.knn(x, q, k)
        assert np.array_equal(idx, z[f"idx{i}"])
        np.testing.assert_allclose(dist, z[f"dist{i}"], rtol=0, atol=1e-13)       # (unit-sized clouds)
        np.testing.assert_allclose(orc.idw_predict(x, y, q, k), z[f"pred{i}"], rtol=1e-10, atol=0)
