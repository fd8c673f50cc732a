"""
The topology engine applies a whole batch of parents on several threads (csrc/topology.cpp, refine_batch_parallel); the
result must be the one the sequential procedure gives -- the reference's numbering is sequential and order dependent
(s_cube.py:904-1536).  Two engines get the same random sequence of batches / relinks / invalid marks, one restricted to the
sequential path, and every table is compared.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
from sparsespatialsampling_amd.s_cube import _Topology
dim, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
t = _Topology(dim, 1.7, np.full(dim, 0.3))
nch = 2 ** dim
leaves = np.array([0], dtype=np.int64)
for lvl in range(3 if dim == 3 else 4):                       # uniform levels (relink pass)
    first = t.submit_refine(leaves, True)
    leaves = np.arange(first, first + nch * len(leaves), dtype=np.int64)
alive = set(leaves.tolist())
for it in range(14):
    pool = np.fromiter(alive, dtype=np.int64)
    pick = rng.permutation(pool)[: int(rng.integers(1, max(2, len(pool) // 6)))]
    t.submit_relink_parent_of(pick[rng.permutation(len(pick))])
    first = t.submit_refine(pick, False)
    new = np.arange(first, first + nch * len(pick), dtype=np.int64)
    alive.difference_update(pick.tolist())
    bad = new[rng.random(len(new)) < 0.04]
    if len(bad):
        t.submit_mark_invalid(bad)
    alive.update(np.setdiff1d(new, bad).tolist())
t.sync()
faces, nodes = t.finalize(np.int64)
np.savez(sys.argv[3], level=t.level, parent=t.parent, first_child=t.first_child, nb=t.nb, node_idx=t.node_idx,
         center=t.center, all_nodes=t.nodes, faces=faces, nodes=nodes)
"""


@pytest.mark.parametrize("dim,seed", [(2, 0), (3, 1), (3, 2)])
def test_parallel_batches_equal_sequential(tmp_path, dim, seed):
    script = tmp_path / "run.py"
    script.write_text(SCRIPT.format(root=ROOT))
    out = {}
    for name, env in (("seq", {"S3_TOPO_THREADS": "1"}), ("par", {"S3_TOPO_THREADS": "4", "S3_TOPO_PAR_MIN": "1"})):
        f = tmp_path / f"{name}.npz"
        subprocess.check_call([sys.executable, str(script), str(dim), str(seed), str(f)], env={**os.environ, **env})
        out[name] = np.load(f)
    assert len(out["seq"]["level"]) > 3000
    for key in out["seq"].files:
        assert np.array_equal(out["seq"][key], out["par"][key]), key
