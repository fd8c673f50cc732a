"""N>1 path on CPU: world_size-2 gloo run of the refine reduction split + shard helpers (no GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch as pt
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p) if p not in sys.path else None
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry, parallel
    from inputs import refine_inputs
    from tests.oracle_backend import OracleTreeBackend
    import logging
    logging.getLogger().setLevel(logging.WARNING)
    s_cube._make_backend = lambda v, t, k: OracleTreeBackend(v, t, k)          # test-only injection (no GPU here)
    assert parallel.world() == (rank, world)
    x, y, geos, kw = refine_inputs("refine_2d_delta", geometry)
    tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
    before = tree._backend.comm.n_collectives
    tree.refine()            # every batch: split KNN work + ONE grouped all-gather; the captured metric needs no exchange
    assert tree._backend.comm.world == world and tree._backend.comm.name == "gloo"
    # SURVEY 8(e) / north_star: one exchange per refinement step -- the all-gather of the batch and nothing else
    assert tree._backend.comm.n_collectives - before == tree._backend.n_batches > 10
    res = dict(metric=np.array(tree._metric), centers=tree.all_centers.numpy(), levels=tree.all_levels.numpy(),
               shard=parallel.shard_range(1001))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        pt.save(gathered, out)
    dist.destroy_process_group()


def test_refine_two_ranks_gloo(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = pt.load(out, weights_only=False)
    z = np.load(os.path.join(ROOT, "tests", "golden", "refine_2d_delta.npz"))
    for r in (r0, r1):                                # both ranks reproduce the reference grid
        assert np.array_equal(r["centers"], z["all_centers"]) and np.array_equal(r["levels"], z["all_levels"])
        np.testing.assert_allclose(r["metric"], z["metric_hist"], rtol=1e-12)
    assert np.array_equal(r0["metric"], r1["metric"])  # identical on all ranks
    assert r0["shard"] == (0, 501) and r1["shard"] == (501, 1001)
    # ... and identical to a single-process run, bit for bit: the reduction order does not depend on the world size
    for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p) if p not in sys.path else None
    import sparsespatialsampling_amd.s_cube as s_cube
    from sparsespatialsampling_amd import geometry
    from inputs import refine_inputs
    from tests.oracle_backend import OracleTreeBackend
    old = s_cube._make_backend
    s_cube._make_backend = lambda v, t, k: OracleTreeBackend(v, t, k)
    try:
        x, y, geos, kw = refine_inputs("refine_2d_delta", geometry)
        tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(y), geometry_obj=geos, **kw)
        tree.refine()
    finally:
        s_cube._make_backend = old
    assert np.array_equal(np.array(tree._metric), r0["metric"])


def test_batch_slice_and_block_shares():
    from sparsespatialsampling_amd.parallel import batch_slice
    for n in (0, 1, 7, 8, 9, 1000, 40001):
        for w in (1, 2, 3, 8):
            parts = [batch_slice(n, r, w) for r in range(w)]
            chunk = parts[0][0]
            assert all(p[0] == chunk for p in parts) and chunk * w >= n
            covered = [i for _, b, e in parts for i in range(b, e)]
            assert covered == list(range(n))
            assert all(b == min(r * chunk, n) for r, (_, b, _e) in enumerate(parts))


def test_shard_range_covers_everything():
    from sparsespatialsampling_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 1000, 461130):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _fallback_worker(rank, world, port, out):
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("S3_DIST_BACKEND", None)
    import logging
    logging.getLogger().setLevel(logging.ERROR)
    from sparsespatialsampling_amd import parallel
    comm = parallel.init()                   # asks for RCCL; there is no GPU here, so no rank gets a communicator
    a = pt.zeros(2 * 5, dtype=pt.float64)
    a[rank * 5:(rank + 1) * 5] = pt.arange(5, dtype=pt.float64) + 10 * rank
    comm.allgather_inplace([a], [5])
    res = dict(name=comm.name, world=(comm.rank, comm.world), gathered=a.numpy().copy(), mx=comm.allreduce_max(float(rank + 1)))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        pt.save(gathered, out)
    parallel.shutdown()
    assert not dist.is_initialized()         # the group the fallback created is the fallback's to destroy


def test_ranks_without_rccl_agree_on_gloo(tmp_path):
    """``parallel.init()`` with two ranks and no usable RCCL communicator (no GPU in the process): the ranks learn through the
    rendezvous store that nobody has one and carry the exchange steps over a gloo group they create themselves"""
    out = str(tmp_path / "res.pt")
    mp.spawn(_fallback_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = pt.load(out, weights_only=False)
    for r, res in enumerate((r0, r1)):
        assert res["name"].startswith("gloo (RCCL") and res["world"] == (r, 2) and res["mx"] == 2.0
        assert np.array_equal(res["gathered"], np.array([0, 1, 2, 3, 4, 10, 11, 12, 13, 14], dtype=np.float64))


def _gather_worker(rank, world, port, out):
    for p in (ROOT,):
        sys.path.insert(0, p) if p not in sys.path else None
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparsespatialsampling_amd import parallel
    comm = parallel.init("gloo")
    counts = [3, 0, 5][:world] if world == 3 else [4, 2]
    send = (pt.arange(counts[rank] * 2, dtype=pt.float64).reshape(counts[rank], 2) + 100 * rank)
    recv = pt.zeros((sum(counts), 2), dtype=pt.float64) if rank == 0 else None
    comm.gather_to_root(send, recv, counts, root=0)           # blocks in rank order on the root, nothing on the others
    prof = pt.zeros((world, 5), dtype=pt.float64)
    prof[rank] = pt.arange(5, dtype=pt.float64) * (rank + 1)
    comm.allgather_inplace([prof], [5])                        # what LeafShards does with its cost profiles
    if rank == 0:
        pt.save(dict(recv=recv, prof=prof, counts=counts), out)
    comm.barrier()
    parallel.shutdown()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_to_root_and_profiles_gloo(tmp_path, world):
    """the exchange of a sharded export (every rank's rows to the rank that writes the file, variable block sizes incl. an
    empty one) and the all-gather of the shards' cost profiles, world_size 2 and 3 on gloo"""
    out = str(tmp_path / "g.pt")
    mp.spawn(_gather_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = pt.load(out, weights_only=False)
    counts = res["counts"]
    want = pt.cat([pt.arange(c * 2, dtype=pt.float64).reshape(c, 2) + 100 * r for r, c in enumerate(counts)])
    assert pt.equal(res["recv"], want)
    assert pt.equal(res["prof"], pt.stack([pt.arange(5, dtype=pt.float64) * (r + 1) for r in range(world)]))


def _shared_worker(rank, world, port, out):
    for p in (ROOT,):
        sys.path.insert(0, p) if p not in sys.path else None
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparsespatialsampling_amd import parallel
    comm = parallel.init("gloo")
    assert comm.broadcast_bytes(b"from the root" if rank == 0 else None) == b"from the root"
    # the hand-over of a sharded export batch: [T, n, n_comp] snapshot-major, every rank fills the rows of its targets
    t, n, n_comp = 3, 17, 2
    if world == 2:
        mine = np.arange(n)[rank::2]                               # interleaved ids
    else:
        mine = [np.array([0, 1, 2, 8, 9]), np.array([], dtype=int), np.array([3, 4, 5, 6, 7] + list(range(10, n)))][rank]
    names = set(os.listdir("/dev/shm"))
    for batch in range(2):                                     # a second buffer: new segment, new name
        shared = parallel.SharedHostArray(comm, t * n * n_comp * 8, register=False)
        assert shared.device_ptr is None and shared.array.nbytes == t * n * n_comp * 8
        view = shared.array.view(np.float64).reshape(t, n, n_comp)
        for i in mine:
            view[:, i, :] = 1000.0 * batch + 10.0 * i + np.arange(t)[:, None] + 0.5 * np.arange(n_comp)[None, :]
        comm.barrier()                                          # everybody's rows are in
        if rank == 0:
            want = 1000.0 * batch + 10.0 * np.arange(n)[None, :, None] + np.arange(t)[:, None, None] + 0.5 * np.arange(n_comp)[None, None, :]
            assert np.array_equal(view, want)
        comm.barrier()
        del view
        shared.close()
    # a /dev/shm that cannot hold the buffer (a container's 64 MB): the root says so, EVERY rank raises -- ExportData then sends
    # the rows to the writing rank through the communicator
    import types
    real_statvfs = os.statvfs
    os.statvfs = lambda path: types.SimpleNamespace(f_bavail=0, f_frsize=4096)
    try:
        try:
            parallel.SharedHostArray(comm, 1 << 20, register=False)
            refused = False
        except parallel.SharedMemoryUnavailable:
            refused = True
    finally:
        os.statvfs = real_statvfs
    assert refused
    leftover = sorted(set(os.listdir("/dev/shm")) - names)
    if rank == 0:
        pt.save(dict(leftover=[f for f in leftover if f.startswith("s3_")]), out)
    comm.barrier()
    parallel.shutdown()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shared_batch_buffer_hand_over_gloo(tmp_path, world):
    """the hand-over of a sharded export: the ranks map ONE host buffer (POSIX shared memory created by
    the root, its name broadcast, unlinked as soon as everybody holds a mapping), every rank writes the rows of its cells --
    interleaved ids, an empty shard -- and the root reads the whole batch in the file's order; no segment is left behind"""
    out = str(tmp_path / "s.pt")
    mp.spawn(_shared_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert pt.load(out, weights_only=False)["leftover"] == []


def test_leaf_shard_cuts_balance_the_cost():
    """``LeafShards._cut``: equal-cost positions from per-stretch cumulative profiles (host logic, no GPU): a curve whose
    second half costs three times the first is cut at 2/3 of its length for two ranks; degenerate profiles keep every rank
    at least one target"""
    from sparsespatialsampling_amd.parallel import LeafShards, shard_range
    sh = LeafShards.__new__(LeafShards)
    sh.world, sh.n = 2, 1200
    first = [shard_range(sh.n, r, 2)[0] for r in range(2)] + [sh.n]
    s = np.arange(LeafShards.PROFILE + 1) / LeafShards.PROFILE
    cuts = sh._cut(np.stack([600.0 * s, 1800.0 * s]), first)
    assert cuts == [0, 800, 1200]                               # 600 + (1200 - 600) / 1800 * 600 = 800
    sh.world, sh.n = 4, 5
    first = [shard_range(5, r, 4)[0] for r in range(4)] + [5]
    assert sh._cut(np.zeros((4, LeafShards.PROFILE + 1)), first) == [0, 2, 3, 4, 5]
    heavy_head = np.zeros((4, LeafShards.PROFILE + 1)); heavy_head[0] = 10.0 * s
    cuts = sh._cut(heavy_head, first)
    assert cuts[0] == 0 and cuts[-1] == 5 and all(b > a for a, b in zip(cuts, cuts[1:]))


def _self_launch(extra_env, args=(), timeout=300):
    import json
    import subprocess
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cylinder3D_small", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"] + list(args), cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    return run, [json.loads(ln) for ln in lines]


def test_self_launched_bench_reports_wedged_ranks_as_an_error_line():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks itself and is their watchdog.  The ranks
    are made to stop before they have touched the GPU, on every backend: the parent kills the attempt, retries once on gloo with FRESH ranks,
    and then prints exactly one JSON line with an `error` field and exits with status 1 -- never a hang, never no line.  (No GPU needed:
    the ranks are stopped before their first HIP call.)"""
    import time
    t0 = time.time()
    run, lines = _self_launch(dict(S3_BENCH_HANG="*:start:any", S3_BENCH_START_TIMEOUT_S="4", S3_BENCH_RUN_TIMEOUT_S="60"))
    assert run.returncode == 1, run.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 2
    att = lines[0]["launcher"]["attempts"]
    assert [a["backend"] for a in att] == ["rccl", "gloo"] and not any(a["ok"] for a in att)
    assert "rank(s) [0, 1] not up" in lines[0]["error"] and "S3_BENCH_HANG" in run.stderr
    assert time.time() - t0 < 120
    assert run.stderr.count("sleeping at stage 'start'") == 4 and run.stderr.count("all ranks killed") == 2      # two attempts x two ranks


def test_self_launch_is_not_taken_under_a_launcher(monkeypatch):
    """with WORLD_SIZE in the environment (torch.distributed.run) bench.py must not start ranks of its own; a mismatch is an
    argument error as before"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=300)
    assert run.returncode == 2 and "starts its own ranks" in run.stderr and not run.stdout.strip()


def test_rccl_bootstrap_timer_ends_a_rank_that_waits_for_ever():
    """``ncclCommInitRank`` has no timeout: ``parallel.init`` arms a timer around the communicator's creation
    (S3_COMM_INIT_TIMEOUT_S) that ends the process with status 86 and a message, so that a launcher sees a FAILED rank instead of a
    silent hang.  Here the creation is replaced by a call that never returns (no GPU needed)."""
    import subprocess
    code = (
        "import sys, time, types; sys.path.insert(0, %r)\n"
        "from sparsespatialsampling_amd import parallel, hipops, _lib\n"
        "hipops.device = lambda: 'cpu'\n"
        "_lib.hip_lib = lambda: types.SimpleNamespace(s3_comm_available=lambda: 1)\n"
        "class Never(parallel.SoloComm):\n"
        "    def __init__(self, *a):\n"
        "        time.sleep(3600)\n"
        "parallel.RcclComm = Never\n"
        "parallel.init()\n"
        "print('returned')\n") % ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", S3_COMM_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               S3_COMM_INIT_TIMEOUT_S="2")
    env.pop("S3_DIST_BACKEND", None)
    run = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 86 and "did not finish within 2 s" in run.stderr and "returned" not in run.stdout, (run.returncode, run.stderr[-1500:])
