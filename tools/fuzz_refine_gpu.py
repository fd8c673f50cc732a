"""dev helper (GPU box): randomised differential test of SamplingTree.refine() with the HIP kernels against the same host
logic driven by the oracle kernels (tests/oracle_backend.py); configurations drawn like tools/fuzz_refine_vs_reference.py
    python tools/fuzz_refine_gpu.py [seed] [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import logging
import types
import numpy as np
import torch as pt

import inputs                                               # tests/golden/inputs.py: the generators the reference fuzz uses
gen = types.SimpleNamespace(bodies=inputs.random_bodies, build=inputs.build_geometries)
from sparsespatialsampling_amd import geometry, s_cube
from oracle_backend import OracleTreeBackend

logging.disable(logging.CRITICAL)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 50
hip_backend = s_cube._make_backend
bad = 0
for case in range(n_cases):
    d = int(rng.integers(2, 4))
    n = int(rng.integers(300, 30000))
    x = rng.random((n, d))
    c0 = rng.random(d)
    y = 0.05 + np.exp(-rng.uniform(2, 12) * np.linalg.norm(x - c0, axis=1)) * (1 + 0.3 * np.sin(9 * x[:, 0]))
    kw = dict(uniform_level=int(rng.integers(1, 5 if d == 2 else 4)))
    if rng.random() < 0.6:
        kw["min_metric"] = float(rng.uniform(0.3, 0.9))
    else:
        kw["n_cells"] = int(rng.integers(100, 6000))
    if rng.random() < 0.3:
        kw["max_delta_level"] = True
    if rng.random() < 0.3:
        kw["n_cells_iter_start"], kw["n_cells_iter_end"] = int(rng.integers(1, 60)), int(rng.integers(1, 10))
    if rng.random() < 0.2:
        kw["pre_select"] = True
    spec = gen.bodies(rng, d)
    trees = []
    for make in (hip_backend, lambda v, t, k: OracleTreeBackend(v, t, k)):
        s_cube._make_backend = make
        try:
            tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=gen.build(geometry, d, spec), **kw)
            tree.refine()
            trees.append(tree)
        except Exception as e:                             # noqa: BLE001
            trees.append(type(e))
    a, b = trees
    tag = dict(case=case, d=d, n=n, kw=kw, bodies=[s[0] for s in spec])
    if isinstance(a, type) or isinstance(b, type):
        ok = a is b
        print(("ok  " if ok else "BAD ") + f"raised {a} / {b}", tag, flush=True)
    else:
        da, db = a._backend.download(a._topo.n_cells), b._backend.download(b._topo.n_cells)
        ok = (np.array_equal(a.all_centers.numpy(), b.all_centers.numpy()) and np.array_equal(a.face_ids.numpy(), b.face_ids.numpy())
              and np.array_equal(a.all_nodes.numpy(), b.all_nodes.numpy()) and list(a._n_cells_log) == list(b._n_cells_log)
              and all(np.array_equal(da[q], db[q]) for q in ("metric", "gain", "center", "level"))
              and np.allclose(np.array(a._metric), np.array(b._metric), rtol=1e-12, atol=0))
        print(("ok  " if ok else "BAD ") + f"{len(a.all_centers)} cells", tag, flush=True)
    bad += not ok
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
