"""
Device-resident topology engine (csrc/topo_dev.hip) against the host engine (csrc/topology.cpp, the
executable specification: pinned table by table against the reference's dumps in test_gpu_refine.py /
test_tree_host_logic.py): random ordered sequences of the engine's operations -- uniform levels, adaptive batches, ordered
relinks with parents and grandparents in one list, removal of clusters of neighbouring cells -- must leave every table
identical, and the final renumbering too.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TABLES = ("level", "parent", "first_child", "nb", "node_idx", "center", "nodes")


def _engines(dim):
    from sparsespatialsampling_amd.s_cube import _DeviceTopology, _Topology
    root = np.array([0.3, -0.2, 0.7][:dim])
    return _Topology(dim, 1.7, root), _DeviceTopology(dim, 1.7, root)


def _same_tables(host, dev, what):
    host.sync(), dev.sync()
    assert host.n_cells == dev.n_cells and host.n_nodes == dev.n_nodes, what
    for name in TABLES:
        a, b = getattr(host, name), getattr(dev, name)
        assert a.shape == b.shape and np.array_equal(a, b), f"{what}: table {name} differs " \
            f"(first at {np.argwhere(np.asarray(a) != np.asarray(b))[:3].tolist()})"


@pytest.mark.parametrize("dim,seed,rounds", [(2, 0, 7), (2, 1, 7), (2, 5, 12), (3, 2, 7), (3, 3, 7), (3, 6, 10), (3, 7, 5)])
def test_device_engine_equals_host_engine_on_random_sequences(dim, seed, rounds):
    rng = np.random.default_rng(seed)
    host, dev = _engines(dim)
    nch = 2 ** dim
    leaves = np.array([0], dtype=np.int64)
    for _ in range(3 if dim == 2 else 2):                       # uniform levels, parents in a scrambled order
        order = rng.permutation(leaves)
        firsts = {e.submit_refine(order, relink=1) for e in (host, dev)}
        assert len(firsts) == 1
        first = firsts.pop()
        leaves = np.arange(first, first + len(order) * nch, dtype=np.int64)
    _same_tables(host, dev, "uniform levels")
    for rnd in range(rounds):
        # a cluster of neighbouring leaves disappears (they list each other: the order of the list decides who is wiped
        # from whose row), plus a few scattered ones
        host.sync()
        nb, fc = host.nb.copy(), host.first_child.copy()
        seeds = rng.choice(leaves, size=min(3, len(leaves)), replace=False)
        gone = set()
        for s0 in seeds:
            gone.add(int(s0))
            gone.update(int(q) for q in nb[s0] if q >= 0 and fc[q] == -1 and rng.random() < 0.7)
        gone = rng.permutation(np.fromiter(gone, dtype=np.int64))
        if len(leaves) - len(gone) > 4:
            for e in (host, dev):
                e.submit_mark_invalid(gone)
            leaves = np.setdiff1d(leaves, gone)
            _same_tables(host, dev, f"round {rnd}: mark_invalid")
        # adaptive batch: a third of the leaves, scrambled; then the ordered relink of a list that holds children of
        # parents AND of grandparents (the refresh of one rewrites the row the other reads), some parents several times
        order = rng.permutation(rng.choice(leaves, size=max(1, len(leaves) // 3), replace=False))
        firsts = {e.submit_refine(order, relink=0) for e in (host, dev)}
        first = firsts.pop()
        leaves = np.concatenate([np.setdiff1d(leaves, order), np.arange(first, first + len(order) * nch, dtype=np.int64)])
        _same_tables(host, dev, f"round {rnd}: refine")
        cells = rng.permutation(rng.choice(leaves, size=max(1, (2 * len(leaves)) // 3), replace=False))
        for e in (host, dev):
            e.submit_relink_parent_of(cells)
        _same_tables(host, dev, f"round {rnd}: relink_parent_of")
    order = np.sort(leaves)
    for dtype in (np.int32, np.int64):
        (f_h, n_h), (f_d, n_d) = host.finalize(dtype), dev.finalize(dtype)
        assert f_h.dtype == f_d.dtype and np.array_equal(f_h, f_d) and np.array_equal(n_h, n_d)
    (c_h, l_h), (c_d, l_d) = host.gather_cells(order[::-1]), dev.gather_cells(order[::-1])
    assert np.array_equal(c_h, c_d) and np.array_equal(l_h, l_d)
    assert dev.check_nb(int(order[-1])) == host.check_nb(int(order[-1]))
    host.close(), dev.close()


def test_device_engine_reports_a_refined_non_leaf():
    _, dev = _engines(2)
    dev.submit_refine(np.array([0]), relink=1)
    dev.sync()
    dev.submit_refine(np.array([0]), relink=0)                   # 0 is a parent now
    with pytest.raises(RuntimeError):
        dev.sync()
    dev.close()


def test_refine_runs_on_the_device_engine_and_equals_the_host_engine(monkeypatch):
    """which engine a tree gets: the device-resident one on the HIP backend, the host engine for the 2:1-balance mode or on
    request (S3_TOPOLOGY=host); both give the same grid, tables and node ids on a golden configuration"""
    import torch as pt
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from inputs import refine_inputs
    from sparsespatialsampling_amd import geometry
    from sparsespatialsampling_amd.s_cube import SamplingTree
    x, y, geos, kw = refine_inputs("refine_3d_metric", geometry)
    results = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("S3_TOPOLOGY", mode)
        tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
        assert type(tree._topo_engine).__name__ == ("_DeviceTopology" if mode == "device" else "_Topology")
        tree.refine()
        topo = tree._topo
        results[mode] = [tree.all_centers.numpy(), tree.all_levels.numpy(), tree.face_ids.numpy(), tree.all_nodes.numpy()] + \
                        [np.array(getattr(topo, name)) for name in TABLES]
        tree.close()
    for a, b in zip(results["device"], results["host"]):
        assert a.shape == b.shape and np.array_equal(a, b)
    z = np.load(os.path.join(ROOT, "tests", "golden", "refine_3d_metric.npz"))
    assert np.array_equal(results["device"][0], z["all_centers"]) and np.array_equal(results["device"][2], z["face_ids"])
    monkeypatch.setenv("S3_TOPOLOGY", "device")
    x2, y2, geos2, kw2 = refine_inputs("refine_3d_delta", geometry)
    tree = SamplingTree(pt.from_numpy(x2), pt.from_numpy(y2), geometry_obj=geos2, **kw2)
    assert type(tree._topo_engine).__name__ == "_Topology"          # max_delta_level=True reads single rows between updates
    tree.close()


def test_device_engine_releases_its_memory():
    """create / refine / finalize / close in a loop: the free device memory (hipMemGetInfo, which sees the engine's own
    hipMalloc allocations) ends where it started"""
    import torch as pt
    from sparsespatialsampling_amd.s_cube import _DeviceTopology
    pt.cuda.synchronize()
    free_before = None
    for it in range(12):
        dev = _DeviceTopology(3, 1.0, np.array([0.5, 0.5, 0.5]))
        leaves = np.array([0], dtype=np.int64)
        for _ in range(5):                                           # 8^5 = 32768 leaves, several table growths
            first = dev.submit_refine(leaves, relink=1)
            leaves = np.arange(first, first + len(leaves) * 8, dtype=np.int64)
        dev.submit_mark_invalid(leaves[:100])
        dev.submit_relink_parent_of(leaves[100:5000])
        faces, nodes = dev.finalize(np.int32)
        assert faces.shape == (len(leaves) - 100, 8)
        dev.close()
        pt.cuda.synchronize()
        if it == 1:                                                  # (the first round also loads code objects / creates pools)
            free_before = pt.cuda.mem_get_info()[0]
    assert abs(pt.cuda.mem_get_info()[0] - free_before) <= (8 << 20), "device memory of closed engines is not released"
