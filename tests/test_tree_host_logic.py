"""
Host logic of SamplingTree (CPython set ordering, native topology engine, stopping rules) against the golden outputs
of the REAL reference.  The numerical kernels are supplied by the CPU oracle through tests/oracle_backend.py (test-only
injection), so this runs without a GPU; tests/test_gpu_refine.py repeats the same comparisons with the HIP backend.
"""
import os

import numpy as np
import pytest
import torch as pt

import sparsespatialsampling_amd.s_cube as s_cube
from sparsespatialsampling_amd import geometry
from inputs import cloud, refine_inputs, sha, wake_metric
from tests.oracle_backend import OracleTreeBackend

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def oracle_backend(monkeypatch):
    monkeypatch.setattr(s_cube, "_make_backend", lambda v, t, k: OracleTreeBackend(v, t, k))


@pytest.fixture
def oracle_grid_backend(monkeypatch):
    monkeypatch.setattr(s_cube, "_make_backend", lambda v, t, k: OracleTreeBackend(v, t, k, grid=True))


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def check_tree_against_golden(tree, z, exact_values=True):
    topo = tree._topo
    n = topo.n_cells
    assert n == len(z["level"])
    assert np.array_equal(topo.level, z["level"])
    assert np.array_equal(topo.parent, z["parent"])
    state = np.where(topo.first_child == -1, 0, np.where(topo.first_child == -2, 2, 1))
    assert np.array_equal(state, z["state"])
    assert np.array_equal(topo.center, z["center"])
    assert np.array_equal(topo.nb, z["nb"])
    assert np.array_equal(topo.node_idx, z["node_idx"])
    assert np.array_equal(np.fromiter(tree._leaf_cells, dtype=np.int64), z["leaf_order"])
    vals = tree._cell_values()
    if exact_values:
        assert np.array_equal(vals["metric"][1:], z["metric"][1:])
        assert np.array_equal(vals["gain"][1:], z["gain"][1:])


def check_outputs_against_golden(tree, z):
    assert np.array_equal(tree.all_centers.numpy(), z["all_centers"])
    assert tree.all_levels.dtype == pt.int64 and np.array_equal(tree.all_levels.numpy(), z["all_levels"])
    assert tree.face_ids.dtype == pt.int32 and np.array_equal(tree.face_ids.numpy(), z["face_ids"])
    assert np.array_equal(tree.all_nodes.numpy(), z["all_nodes"])
    assert np.array_equal(np.array(tree._n_cells_log), z["n_cells_log"])
    # captured-metric history: the only non bit-exact quantity (summation order of ||.||_2), see DESIGN.md
    np.testing.assert_allclose(np.array(tree._metric), z["metric_hist"], rtol=1e-12, atol=0)
    info = tree.data_final_mesh
    assert info["iterations"] == int(z["iterations"])
    assert info["min_level"] == int(z["min_level"]) and info["max_level"] == int(z["max_level"])
    assert info["n_cells"] == len(z["all_levels"])


@pytest.mark.parametrize("d", [2, 3])
def test_uniform_tables(oracle_backend, d):
    z = load(f"uniform_{d}d")
    x = cloud(61 + d, 500, [0.0] * d, [10.0] * d)
    y = wake_metric(x / 10.0, [0.3, 0.4, 0.0][:d])
    assert np.array_equal(x, z["x"]) and np.array_equal(y, z["y"])
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), uniform_level=3,
                               geometry_obj=[geometry.CubeGeometry("domain", True, [0] * d, [10] * d)])
    tree._refine_uniform()
    check_tree_against_golden(tree, z)
    assert np.array_equal(tree._topo.nodes, z["all_nodes"])
    assert tree._topo.n_nodes == (2 ** 3 + 1) ** d            # perfect sharing on a uniform grid: 81 / 729


@pytest.mark.parametrize("name", ["refine_2d_metric", "refine_2d_ncells", "refine_2d_delta", "refine_3d_metric",
                                  "refine_3d_delta", "refine_3d_ncells_cone", "refine_2d_triangle", "refine_3d_polytopes",
                                  "refine_2d_polygon"])
def test_refine_matches_reference(oracle_backend, name):
    z = load(name)
    x, y, geos, kw = refine_inputs(name, geometry)
    assert sha(x, y) == str(z["input_sha"])
    trace = []
    orig = s_cube.SamplingTree._refine_cells

    def traced(self, to_refine):
        trace.append(np.fromiter(to_refine, dtype=np.int64))
        return orig(self, to_refine)

    s_cube.SamplingTree._refine_cells = traced
    try:
        tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
        tree.refine()
    finally:
        s_cube.SamplingTree._refine_cells = orig
    assert float(tree._width) == float(z["width"]) and tree._gain0 == float(z["gain0"])
    # which cells were split, in which order, in every iteration (incl. geometry refinement)
    assert np.array_equal(np.array([len(t) for t in trace]), z["trace_len"])
    assert np.array_equal(np.concatenate(trace), z["trace"])
    check_tree_against_golden(tree, z)
    check_outputs_against_golden(tree, z)


RANDOM_SEEDS = [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11]        # seed 3: the reference itself raises (single-cell iteration)


def check_random_case(seed):
    """randomly drawn configuration ``seed`` (tests/golden/inputs.py: random_refine_case) against the real reference"""
    from inputs import build_geometries, random_refine_case
    z = load(f"refine_random_{seed}")
    x, y, spec, kw, d = random_refine_case(seed)
    assert sha(x, y) == str(z["input_sha"])
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=build_geometries(geometry, d, spec), **kw)
    tree.refine()
    assert np.array_equal(tree.all_centers.numpy(), z["all_centers"])
    assert np.array_equal(tree.all_levels.numpy(), z["all_levels"].astype(np.int64))
    assert np.array_equal(tree.face_ids.numpy(), z["face_ids"])
    assert np.array_equal(tree.all_nodes.numpy(), z["all_nodes"])
    assert np.array_equal(np.array(tree._n_cells_log), z["n_cells_log"])
    np.testing.assert_allclose(np.array(tree._metric), z["metric_hist"], rtol=1e-12, atol=0)
    assert tree.data_final_mesh["iterations"] == int(z["iterations"])


@pytest.mark.parametrize("seed", RANDOM_SEEDS)
def test_refine_random_configurations(oracle_backend, seed):
    check_random_case(seed)


def test_c1_cylinder2d_full_size(oracle_backend):
    """BASELINE config C1 at full size, host logic + oracle kernels vs the real reference (87 adaptive iterations)"""
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


@pytest.mark.parametrize("seed", [0, 5, 9])
def test_refine_random_configurations_grid_oracle(oracle_grid_backend, seed):
    """the same reference grids with the oracle's bucket-grid neighbour search (what bench.py times as the CPU port)"""
    check_random_case(seed)


def test_close_twice_is_harmless(oracle_backend):
    """``SamplingTree.close()`` is an optional early release: calling it again (or after an exception handler already did)
    must not raise -- also on the host topology engine (2:1 balance mode), whose ``sync`` used to insist on a live handle"""
    x, y, geos, kw = refine_inputs("refine_2d_delta", geometry)
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
    tree.refine()
    centers = tree.all_centers.clone()
    tree.close()
    tree.close()
    assert pt.equal(tree.all_centers, centers)                # the results stay valid


class DuckGeometry:
    """a user-defined geometry with the reference's interface only (s_cube.py:1816-1837 calls nothing but ``check_cell``):
    no ``kernel_spec``, every verdict comes from the wrapped object's host predicate"""

    def __init__(self, inner):
        self._inner = inner

    def check_cell(self, cell_nodes, refine_geometry=False):
        return self._inner.check_cell(cell_nodes, refine_geometry)

    def __getattr__(self, item):
        if item == "kernel_spec":
            raise AttributeError(item)
        return getattr(self._inner, item)


def check_geometry_fallback(name="refine_2d_metric"):
    """bodies WITHOUT a device predicate take the host path (check_cell on the cells' node coordinates) and give the
    reference's grid all the same"""
    z = load(name)
    x, y, geos, kw = refine_inputs(name, geometry)
    geos = [g if g.keep_inside else DuckGeometry(g) for g in geos]
    assert any(isinstance(g, DuckGeometry) for g in geos)
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
    tree.refine()
    check_tree_against_golden(tree, z)
    check_outputs_against_golden(tree, z)


@pytest.mark.parametrize("name", ["refine_2d_metric", "refine_3d_metric"])
def test_geometry_without_kernel_spec_takes_the_host_path(oracle_backend, name):
    check_geometry_fallback(name)


def test_base_class_kernel_spec_defaults_to_none():
    class Slab(geometry.GeometryObject):
        type, main_width, center = "slab", 1.0, pt.zeros(2)

        def check_cell(self, cell_nodes, refine_geometry=False):
            return self._apply_mask(cell_nodes[:, 0] <= 0.5, refine_geometry)

        def _compute_main_width(self):
            return 1.0

        def _compute_center(self):
            return pt.zeros(2)

    g = Slab("slab", False)
    assert g.kernel_spec() is None
    assert g.check_cell(pt.tensor([[0.1, 0.0], [0.2, 0.0], [0.3, 1.0], [0.4, 1.0]])) is True


def test_bench_grid_sha_and_cpu_counts():
    """helpers of bench.py that need no GPU: the grid digest depends on values, dtypes and shapes of all four arrays (what two
    backends must agree on cell for cell), and the host-core report has the three counts north_star asks to be stated"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root) if root not in sys.path else None
    import numpy as np
    import torch as pt
    import bench
    c = pt.arange(12, dtype=pt.float64).reshape(4, 3)
    lv = pt.ones((4, 1), dtype=pt.int64)
    f = pt.arange(32, dtype=pt.int32).reshape(4, 8)
    nd = pt.arange(30, dtype=pt.float64).reshape(10, 3)
    ref = bench.grid_sha(c, lv, f, nd)
    assert ref == bench.grid_sha(c.clone(), lv.numpy(), f.clone(), nd.numpy()) and len(ref) == 64
    assert ref != bench.grid_sha(c, lv, f.long(), nd)                          # dtype is part of the contract (faces int32 / int64)
    bumped = c.clone()
    bumped[3, 2] = np.nextafter(float(bumped[3, 2]), 1e9)
    assert ref != bench.grid_sha(bumped, lv, f, nd)                            # one ulp in one centre
    assert ref != bench.grid_sha(c, lv, f, nd.reshape(3, 10))
    counts = bench.host_cpu_counts()
    assert set(counts) == {"host_cpus", "affinity_cpus", "cgroup_cpus"} and counts["host_cpus"] >= counts["affinity_cpus"] >= 1
    assert counts["cgroup_cpus"] is None or counts["cgroup_cpus"] > 0
