"""
Golden-vector generator: runs the REAL reference (read-only at /root/reference, imported through the shims in
``ref_stubs.py``) on small seeded inputs and stores inputs/outputs as ``.npz`` fixtures next to this script.

Run in the development container only (the reference never travels to the GPU box):

    python tests/golden/gen_golden.py            # all fixtures
    python tests/golden/gen_golden.py interp     # one group

Fixture groups:
    interp_*      export.interpolate_data                        (reference export.py:446-468)
    knncache_*    ExportData._build_knn_cache idx / weights      (reference export.py:403-444)
    predict_*     KNeighborsRegressor(weights="distance").predict as used at s_cube.py:161-163,224,328,372
    masks         geometry check_cell truth tables                (geometry/*.py)
    masks_polytopes   the same for triangle / prism / tetrahedron / pyramid
    uniform_*     _refine_uniform() neighbour + node tables       (s_cube.py:508-561, 904-1536)
    refine_*      full SamplingTree.refine() outputs + traces     (s_cube.py:563-667)
    refine_random_<seed>   final grids of randomly drawn configurations (inputs.random_refine_case)
    export_*      the REAL ExportData.export / Datawriter / XDMFWriter (export.py:128-319, data.py:303-777) driven through
                  inputs.run_export_case; h5py = h5py_standin.py (ctypes on the HDF5 C library).  Stored: the inventory of
                  every file written (path -> array, hence shape and dtype) and the XDMF text

The script must stay a real file with a ``__main__`` guard: the reference creates a *spawn* multiprocessing pool
(s_cube.py:159) whose workers re-import this module and need the shims installed before unpickling.
"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402,F401  (installs shims + sys.path; must run in spawned workers too)

import numpy as np  # noqa: E402
import torch as pt  # noqa: E402

from inputs import (EXPORT_CASES, POLYTOPES, REFINE_CASES, build_geometries, c1_cylinder2d, cloud, mask_cells, polytope,  # noqa: E402
                    polytope_cells, random_refine_case, refine_inputs, run_export_case, sha, wake_metric)


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------------------------------------------
def gen_interp():
    from sparseSpatialSampling.export import interpolate_data
    for tag, (n, nc, k, ncomp, t, dtype) in {
        "interp_k8_c1_f32": (800, 300, 8, 1, 7, np.float32),
        "interp_k8_c3_f64": (800, 300, 8, 3, 7, np.float64),
        "interp_k26_c1_f32": (800, 300, 26, 1, 7, np.float32),
        "interp_k26_c3_f32": (800, 300, 26, 3, 5, np.float32),
        "interp_k26_c1_f64": (800, 300, 26, 1, 33, np.float64),
    }.items():
        rng = np.random.default_rng(abs(hash(tag)) % 2 ** 31 if False else sum(map(ord, tag)))
        w = rng.random((nc, k))
        w /= w.sum(1, keepdims=True)
        idx = rng.integers(0, n, (nc, k), dtype=np.int64)
        data = rng.standard_normal((n, ncomp, t)).astype(dtype)
        out = interpolate_data(pt.from_numpy(w), pt.from_numpy(idx), pt.from_numpy(data), chunk_size=128)
        save(tag, w=w, idx=idx, data=data, out=out.numpy())


def gen_knncache():
    from sklearn.neighbors import NearestNeighbors
    from sparseSpatialSampling.export import ExportData

    class _FakeSCube:  # the attributes ExportData.__init__ reads (export.py:74-83)
        def __init__(self, centers, vertices):
            self.n_dimensions = centers.shape[1]
            self.faces = None
            self.centers = pt.from_numpy(centers)
            self.vertices = pt.from_numpy(vertices)
            self.levels = None
            self.metric = None
            self.size_initial_cell = 1.0
            self.save_path, self.save_name, self.grid_name = "/tmp", "x", "g"

    for tag, (d, n, nc, nv) in {"knncache_2d": (2, 3000, 400, 120), "knncache_3d": (3, 4000, 300, 90)}.items():
        coords = cloud(11 + d, n, [0.0] * d, [1.0, 0.7, 0.4][:d])
        centers = cloud(21 + d, nc, [0.0] * d, [1.0, 0.7, 0.4][:d])
        vertices = cloud(31 + d, nv, [-0.1] * d, [1.1, 0.8, 0.5][:d])     # some queries outside the cloud
        centers[5] = coords[17]                                            # exact hit -> clamp(1e-12) path
        centers[6] = coords[n - 1]
        ex = ExportData(_FakeSCube(centers, vertices), write_times=["0"], interpolate_at_vertices=True, n_jobs=1)
        ex._build_knn_cache(pt.from_numpy(coords))
        nn = NearestNeighbors(n_neighbors=8 if d == 2 else 26).fit(coords)
        dist_c, _ = nn.kneighbors(centers)
        save(tag, coords=coords, centers=centers, vertices=vertices,
             idx_c=ex._knn_idx_centers.numpy(), w_c=ex._knn_w_centers.numpy(), dist_c=dist_c,
             idx_v=ex._knn_idx_vertices.numpy(), w_v=ex._knn_w_vertices.numpy())


def gen_predict():
    from sklearn.neighbors import KNeighborsRegressor
    for tag, (d, n, nq, k) in {"predict_2d": (2, 2500, 500, 8), "predict_3d": (3, 3000, 500, 26)}.items():
        x = cloud(41 + d, n, [0.0] * d, [2.2, 0.41, 0.3][:d])
        y = wake_metric(x, [0.2, 0.2, 0.0][:d])
        q = cloud(51 + d, nq, [-0.05] * d, [2.25, 0.45, 0.35][:d])
        q[3] = x[10]                                                       # exact hit: indicator weights
        q[4] = x[11]
        knn = KNeighborsRegressor(n_neighbors=k, weights="distance", n_jobs=1).fit(x, y)
        save(tag, x=x, y=y, q=q, pred=knn.predict(q))


def gen_masks():
    from sparseSpatialSampling.geometry import CubeGeometry, SphereGeometry, CylinderGeometry3D, \
        GeometryCoordinates2D
    out = {}
    rng = np.random.default_rng(7)

    def cells(d, n):
        c, h = mask_cells(d, n, rng)
        dirs = np.array([[-1, -1], [-1, 1], [1, 1], [1, -1]] if d == 2 else
                        [[-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1],
                         [-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1]], dtype=np.float64)
        return c[:, None, :] + dirs[None] * h[:, None, None]              # [n, 2^d, d]

    c2, c3 = cells(2, 400), cells(3, 400)
    out["cells2"], out["cells3"] = c2, c3
    poly = [(0.1, 0.1), (0.9, 0.2), (1.2, 0.8), (0.6, 0.5), (0.2, 1.0)]   # concave pentagon
    out["poly"] = np.asarray(poly)
    geos = {}
    for ki in (True, False):
        geos[f"cube2_{int(ki)}"] = (CubeGeometry("g", ki, [0.0, 0.1], [1.0, 0.9]), c2)
        geos[f"cube3_{int(ki)}"] = (CubeGeometry("g", ki, [0.0, 0.1, -0.2], [1.0, 0.9, 0.7]), c3)
        geos[f"sphere2_{int(ki)}"] = (SphereGeometry("g", ki, [0.4, 0.5], 0.45), c2)
        geos[f"sphere3_{int(ki)}"] = (SphereGeometry("g", ki, [0.4, 0.5, 0.3], 0.6), c3)
        geos[f"cyl3_{int(ki)}"] = (CylinderGeometry3D("g", ki, [(0.2, 0.3, -0.1), (0.9, 0.6, 0.8)], 0.35), c3)
        geos[f"cone3_{int(ki)}"] = (CylinderGeometry3D("g", ki, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], [0.5, 0.1]), c3)
        geos[f"poly2_{int(ki)}"] = (GeometryCoordinates2D("g", ki, poly), c2)
    for key, (g, c) in geos.items():
        for rm in (False, True):
            out[f"{key}_r{int(rm)}"] = np.array([g.check_cell(pt.from_numpy(c[i]), rm) for i in range(len(c))])
    save("masks", **out)


def gen_masks_polytopes():
    """check_cell truth tables of TriangleGeometry / PrismGeometry3D / TetrahedronGeometry3D / PyramidGeometry3D"""
    from sparseSpatialSampling import geometry
    rng = np.random.default_rng(11)
    out = {}
    dirs = {2: np.array([[-1, -1], [-1, 1], [1, 1], [1, -1]], dtype=np.float64),
            3: np.array([[-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1],
                         [-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1]], dtype=np.float64)}
    for d in (2, 3):
        c, h = polytope_cells(d, rng)
        out[f"c{d}"], out[f"h{d}"] = c, h
    for key in POLYTOPES:
        d = 2 if key.startswith("tri") else 3
        nodes = out[f"c{d}"][:, None, :] + dirs[d][None] * out[f"h{d}"][:, None, None]
        for ki in (True, False):
            g = polytope(geometry, key, ki)
            for rm in (False, True):
                out[f"{key}_{int(ki)}_r{int(rm)}"] = np.array([g.check_cell(pt.from_numpy(nodes[i]), rm)
                                                               for i in range(len(nodes))])
    save("masks_polytopes", **out)


def _tree_arrays(tree):
    """Dump the per-cell state of a reference SamplingTree into arrays (None -> -1)."""
    cells = tree._cells
    nd = tree.n_dimensions
    nb = np.array([[(-1 if n is None else n.index) for n in c.nb] for c in cells], dtype=np.int32)
    node_idx = np.array([c.node_idx for c in cells], dtype=np.int64)
    level = np.array([c.level for c in cells], dtype=np.int32)
    state = np.array([0 if c.children is None else (2 if len(c.children) == 0 else 1) for c in cells], dtype=np.int8)
    parent = np.array([-1 if c.parent is None else c.parent.index for c in cells], dtype=np.int32)
    center = np.stack([np.asarray(c.center, dtype=np.float64).reshape(nd) for c in cells])
    gain = np.array([float(c.gain) for c in cells], dtype=np.float64)
    metric = np.array([float(c.metric) for c in cells], dtype=np.float64)
    return dict(nb=nb, node_idx=node_idx, level=level, state=state, parent=parent, center=center, gain=gain,
                metric=metric, leaf_order=np.array(list(tree._leaf_cells), dtype=np.int64))


def gen_uniform():
    from sparseSpatialSampling.geometry import CubeGeometry
    from sparseSpatialSampling.s_cube import SamplingTree
    for d in (2, 3):
        x = cloud(61 + d, 500, [0.0] * d, [10.0] * d)
        y = wake_metric(x / 10.0, [0.3, 0.4, 0.0][:d])
        tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(y), uniform_level=3,
                            geometry_obj=[CubeGeometry("domain", True, [0] * d, [10] * d)])
        tree._refine_uniform()
        tree._pool.close()
        arr = _tree_arrays(tree)
        save(f"uniform_{d}d", x=x, y=y, all_nodes=pt.stack(tree.all_nodes).numpy(), **arr)


def gen_refine(only=None):
    from sparseSpatialSampling.s_cube import SamplingTree
    for tag, case in REFINE_CASES.items():
        if only and tag not in only:
            continue
        import sparseSpatialSampling.geometry as ref_geometry
        x, y, geos, kw = refine_inputs(tag, ref_geometry)
        trace = []
        orig = SamplingTree._refine_cells

        def _traced(self, to_refine, _orig=orig, _trace=trace):
            _trace.append(np.array(list(to_refine), dtype=np.int64))
            return _orig(self, to_refine)

        SamplingTree._refine_cells = _traced
        try:
            tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, n_jobs=2, **kw)
            tree.refine()
        finally:
            SamplingTree._refine_cells = orig
        arr = _tree_arrays(tree)
        info = tree.data_final_mesh
        save(tag, input_sha=np.array(sha(x, y)), n_points=np.array(len(x)),
             all_centers=tree.all_centers.numpy(), all_levels=tree.all_levels.numpy(),
             face_ids=tree.face_ids.numpy(), all_nodes=tree.all_nodes.numpy(),
             metric_hist=np.array(tree._metric), n_cells_log=np.array(tree._n_cells_log),
             trace_len=np.array([len(t) for t in trace]), trace=np.concatenate(trace) if trace else np.zeros(0),
             width=np.array(float(tree._width)), gain0=np.array(float(tree._cells[0].gain)),
             iterations=np.array(info["iterations"]), min_level=np.array(info["min_level"]),
             max_level=np.array(info["max_level"]), **arr)


def gen_refine_random():
    """randomly drawn configurations (inputs.random_refine_case): final grid + histories; seeds the reference cannot
    finish (its single-cell-iteration IndexError) are skipped and listed"""
    import sparseSpatialSampling.geometry as ref_geometry
    from sparseSpatialSampling.s_cube import SamplingTree
    done = []
    for seed in range(12):
        x, y, spec, kw, d = random_refine_case(seed)
        try:
            tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=build_geometries(ref_geometry, d, spec),
                                n_jobs=2, **kw)
            tree.refine()
        except IndexError:
            print(f"seed {seed}: reference raised IndexError, skipped")
            continue
        done.append(seed)
        save(f"refine_random_{seed}", input_sha=np.array(sha(x, y)), all_centers=tree.all_centers.numpy(),
             all_levels=tree.all_levels.numpy().astype(np.int8), face_ids=tree.face_ids.numpy(),
             all_nodes=tree.all_nodes.numpy(), metric_hist=np.array(tree._metric), n_cells_log=np.array(tree._n_cells_log),
             iterations=np.array(tree.data_final_mesh["iterations"]))
    print("seeds with a fixture:", done)


def gen_c1():
    """full-size C1 run of the reference (87 adaptive iterations): grid + histories only"""
    import sparseSpatialSampling.geometry as ref_geometry
    from sparseSpatialSampling.s_cube import SamplingTree
    x, m, geos, kw = c1_cylinder2d(ref_geometry)
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(m), geometry_obj=geos, n_jobs=4, **kw)
    tree.refine()
    save("c1_cylinder2d", input_sha=np.array(sha(x, m)), all_centers=tree.all_centers.numpy(),
         all_levels=tree.all_levels.numpy().astype(np.int8), face_ids=tree.face_ids.numpy(),
         metric_hist=np.array(tree._metric), n_cells_log=np.array(tree._n_cells_log),
         iterations=np.array(tree.data_final_mesh["iterations"]))


def gen_c2():
    """full-size C2 run of the reference (OAT15-like: 3*10^5 clustered points, refined NACA outline, n_cells_max): metric =
    std_t(p) + std_t(|U|) over 2000 synthetic snapshots, computed the way the reference's scripts do (torch on the CPU,
    examples/s3_for_OAT15_airfoil.py:91), then rounded to float16 and stored; grid as checksums + histories.  The polygon
    predicate the reference gets here is ref_stubs._Polygon (shapely is absent): GEOS semantics are NOT pinned by this."""
    import sparseSpatialSampling.geometry as ref_geometry
    from sparseSpatialSampling.s_cube import SamplingTree
    from inputs import c2_cloud, c2_fields, c2_oat15
    x, _ = c2_cloud()
    n_snap, step = 2000, 100
    # streaming mean / M2 over snapshot blocks in float64 = torch.std(field, dim=-1) up to rounding
    cnt, mean_p, m2_p, mean_u, m2_u = 0, 0.0, 0.0, 0.0, 0.0
    for t0 in range(0, n_snap, step):
        p, u = c2_fields(x, t0, t0 + step)
        p = pt.from_numpy(p)[:, 0, :].double()
        un = pt.from_numpy(u).double().norm(dim=1)
        for name, blk in (("p", p), ("u", un)):
            b_mean, b_m2, b_n = blk.mean(-1), ((blk - blk.mean(-1, keepdim=True)) ** 2).sum(-1), blk.shape[-1]
            mean, m2 = (mean_p, m2_p) if name == "p" else (mean_u, m2_u)
            delta = b_mean - mean
            tot = cnt + b_n
            m2 = m2 + b_m2 + delta ** 2 * cnt * b_n / tot
            mean = mean + delta * b_n / tot
            if name == "p":
                mean_p, m2_p = mean, m2
            else:
                mean_u, m2_u = mean, m2
        cnt += step
    metric = (pt.sqrt(m2_p / (cnt - 1)) + pt.sqrt(m2_u / (cnt - 1))).numpy().astype(np.float16)
    x, m, geos, kw = c2_oat15(ref_geometry, metric)
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(m), geometry_obj=geos, n_jobs=8, **kw)
    tree.refine()
    c, lv, f, nd = tree.all_centers.numpy(), tree.all_levels.numpy(), tree.face_ids.numpy(), tree.all_nodes.numpy()
    save("c2_oat15", metric_f16=metric, input_sha=np.array(sha(x, m)), n_leaf=np.array(len(c)),
         sha_centers=np.array(sha(c)), sha_levels=np.array(sha(lv.astype(np.int64))), sha_faces=np.array(sha(f.astype(np.int32))),
         sha_nodes=np.array(sha(nd)), head_centers=c[:64], head_faces=f[:64], level_hist=np.bincount(lv.reshape(-1)),
         metric_hist=np.array(tree._metric), n_cells_log=np.array(tree._n_cells_log),
         iterations=np.array(tree.data_final_mesh["iterations"]))
    print("C2:", len(x), "points ->", len(c), "cells, levels", np.bincount(lv.reshape(-1)))


def h5_inventory(path):
    """[(dataset path, array)] of a whole HDF5 file in h5py's iteration order (name order), read with the stand-in"""
    import h5py
    out = []

    def walk(group, prefix):
        for k in group.keys():
            node = group[k]
            if hasattr(node, "keys"):
                walk(node, f"{prefix}{k}/")
            else:
                out.append((f"{prefix}{k}", np.asarray(node[()])))
    with h5py.File(path, "r") as f:
        walk(f, "")
    return out


def gen_export():
    """the reference's export state machine, writer and XDMF writer on small grids the reference generated itself"""
    import json
    import shutil
    import tempfile
    import sparseSpatialSampling.geometry as ref_geometry
    from sparseSpatialSampling.export import ExportData
    from sparseSpatialSampling.sparse_spatial_sampling import SparseSpatialSampling
    for tag, case in EXPORT_CASES.items():
        x, y, geos, kw = refine_inputs(case["refine"], ref_geometry)
        kw = {{"uniform_level": "uniform_levels", "n_cells": "n_cells_max"}.get(k, k): v for k, v in kw.items()}
        tmp = tempfile.mkdtemp(prefix="s3_export_golden_")
        try:
            s3 = SparseSpatialSampling(pt.from_numpy(x), pt.from_numpy(y), geos, tmp, "case", n_jobs=2, **kw)
            s3.execute_grid_generation()
            info = run_export_case(tag, s3, ExportData, pt.from_numpy, x, y)
            arrays, manifest, xdmf = {}, [], {}
            for fn in sorted(os.listdir(tmp)):
                if fn.endswith(".h5"):
                    for path, a in h5_inventory(os.path.join(tmp, fn)):
                        arrays[f"a{len(manifest)}"] = a
                        manifest.append([fn, path, str(a.dtype), list(a.shape)])
                elif fn.endswith(".xdmf"):
                    xdmf[fn] = open(os.path.join(tmp, fn)).read()
            save(tag, input_sha=np.array(sha(x, y)), manifest=np.array(json.dumps(manifest)), xdmf=np.array(json.dumps(xdmf)),
                 error_second_file=np.array(info["error_second_file"]), **arrays)
            print(f"{tag}: {len(manifest)} datasets in {sorted({m[0] for m in manifest})}, xdmf {sorted(xdmf)}, "
                  f"second-file error {info['error_second_file']!r}")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    groups = sys.argv[1:] or ["interp", "knncache", "predict", "masks", "masks_polytopes", "uniform", "refine", "refine_random", "c1", "export"]
    pt.manual_seed(0)
    for g in groups:
        if g.startswith("refine_") and g != "refine_random":
            gen_refine(only=[g])
        else:
            globals()["gen_" + g]()
