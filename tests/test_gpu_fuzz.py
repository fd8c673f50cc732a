"""
Short runs of the randomised differential tools (tools/fuzz_*.py) as part of the GPU suite: planned vs direct
interpolation kernel, bucket-grid KNN vs the oracle's brute force, ExportData end to end vs the oracle, refine() with the
HIP kernels vs the same host logic on the oracle kernels.  Each tool runs once, in its own process, one after the other.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,seed,cases", [("fuzz_interp.py", 21, 60), ("fuzz_knn.py", 22, 60), ("fuzz_export.py", 23, 30),
                                             ("fuzz_refine_gpu.py", 24, 12)])
def test_fuzz_tool(tool, seed, cases):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seed), str(cases)], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    tail = "\n".join((run.stdout + run.stderr).splitlines()[-15:])
    assert run.returncode == 0 and f"{cases} cases, 0 mismatches" in run.stdout, tail


@pytest.mark.gpu
def test_bench_contract_small_workload():
    """bench.py on the reduced workload: exactly one line on stdout, valid JSON, the keys of the driver's contract plus the
    roofline and cpu_baseline objects"""
    import json
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cylinder3D_small", "--steps", "3",
                          "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = run.stdout.strip().splitlines()
    assert len(lines) == 1, run.stdout[-2000:]
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "end_to_end", "refine_cpu_baseline",
                "refine_wall_s"):
        assert key in res, key
    assert res["n_gpus"] == 1 and res["steps"] == 3 and res["warmup"] == 1 and res["higher_is_better"] is True
    assert res["scaling"] == "strong" and res["vs_baseline"] is None and res["data"] == "synthetic" and "workload" in res["config"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(res["roofline"])
    # the traffic of the dominant kernel is MEASURED in the run (two rocprofv3 --pmc child passes before the GPU is touched), where a
    # profiler is installed: its own source line says so, and it cannot be below what the launch must move at least once
    import shutil
    if shutil.which("rocprofv3"):
        m = res["roofline"]["traffic_measured"]
        assert m is not None and m["launches_averaged"] == [3, 3] and "measured in this run" in res["roofline"]["traffic_source"]
        assert 0.9 < res["roofline"]["traffic_over_algorithmic"] < 3 and res["roofline"]["traffic_stale"] is False
        # ... and so is the traffic of every batch shape printed (the same child passes launch them behind the headline): no
        # `traffic_stale` anywhere in the line, no figure taken over from an earlier collection
        assert set(m["batches"]) == {f"{n}/{w}" for n in ("T25", "T25x3", "T100") for w in ("inplace", "pitched")}
        for name, rec in res["roofline_batches"].items():
            assert rec["traffic_stale"] is False and "measured in this run" in rec["traffic_source"], name
            assert 0.9 < rec["traffic_over_algorithmic"] < 4 and 0.9 < rec["pitched_copy"]["traffic_over_algorithmic"] < 4, name
    assert "mix_7r2w" in res["roofline"]["yardsticks"] and res["roofline"]["yardsticks"]["mix_7r2w"]["GBs"] > 1000
    if res["roofline"].get("traffic") is not None:
        assert 0.05 < res["roofline"]["frac_of_streaming_ceiling"] < 1.2
    assert res["roofline"]["bound"] == "hbm" and 0 < res["roofline"]["frac"] < 1 and res["value"] > 0
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(res["cpu_baseline"]) and res["cpu_baseline"]["kind"] == "port"
    assert res["config"]["parallelism"] == "leaf-cell shards x1"
    assert res["refine_cpu_baseline"]["same_grid_size"] and res["refine_cpu_baseline"]["speedup"] > 1
    # full-size parity in the bench run itself: centres, levels, faces, nodes of the CPU port's grid == the HIP backend's, by SHA-256
    assert res["refine_cpu_baseline"]["same_grid_sha"] and res["refine_cpu_baseline"]["grid_sha256_gpu"] == res["grid_sha256"]
    assert set(("host_cpus", "cgroup_cpus", "threads_used", "all_cells")) <= set(res["cpu_baseline"]) and res["cpu_baseline"]["all_cells"]
    assert res["end_to_end"]["T25"]["Gcells_snapshots_per_s"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    """rehearsal of the N = 2 path on one GPU (two processes, gloo for the exchange steps): the grid and the captured
    metric must be the ones of the single-rank run (rank-count independent reduction), the bench line says leaf-cell
    shards x2"""
    import json
    env = dict(os.environ, S3_BENCH_SHARE_GPU="1", S3_DIST_BACKEND="gloo")
    base = [os.path.join(ROOT, "bench.py"), "--workload", "cylinder3D_small", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable] + base, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29611"] + base + ["--gpus", "2"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    r1, r2 = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(two.stdout.strip().splitlines()[-1])
    assert r2["n_gpus"] == 2 and r2["config"]["parallelism"] == "leaf-cell shards x2" and r2["config"]["collectives"] == "gloo"
    assert r1["config"]["n_cells"] == r2["config"]["n_cells"] and r1["refine_cells_created"] == r2["refine_cells_created"]
    assert r1["captured_metric"] == r2["captured_metric"] and r1["refine_iterations"] == r2["refine_iterations"]


def _self_launched_bench(extra_env, timeout=900, gpus=2):
    import json
    env = dict(os.environ, S3_BENCH_SHARE_GPU="1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "S3_DIST_BACKEND"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", "cylinder3D_small", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return run, [json.loads(ln) for ln in run.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_five_ranks_share_one_gpu():
    """the largest world size one GPU box allows next to the test process itself (the pool's guard: six processes on the card): five
    self-launched ranks, an ODD number -- uneven leaf-cell shards, uneven batch slices, five mappings of the shared batch buffer --
    give one line, the single-rank grid and captured metric, and no collective on the export's data path.  (Eight ranks are
    rehearsed on the CPU: tests/test_parallel_gloo.py.)"""
    run, lines = _self_launched_bench({}, gpus=5)
    assert run.returncode == 0, run.stderr[-3000:]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 5 and line["value"] > 0 and line["config"]["parallelism"] == "leaf-cell shards x5"
    assert len(line["config"]["cells_per_rank"]) == 5 and sum(line["config"]["cells_per_rank"]) == line["config"]["n_cells"]
    assert min(line["config"]["cells_per_rank"]) > 0
    assert line["export_sharded"]["data_path_collectives_per_batch"] == 0 and "ms_per_batch" in line["export_sharded"]
    import json
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cylinder3D_small", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert r1["config"]["n_cells"] == line["config"]["n_cells"] and r1["captured_metric"] == line["captured_metric"]
    assert r1["grid_sha256"] == line["grid_sha256"]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` the way the driver runs `--gpus 1` -- no torch.distributed.run around it: the parent (which never
    touches the GPU) starts two fresh ranks, watches them and relays rank 0's ONE JSON line.  Two ranks share the one GPU here; RCCL
    refuses that inside the ranks, which agree on gloo among themselves (parallel.init) -- the first attempt succeeds."""
    run, lines = _self_launched_bench({})
    assert run.returncode == 0, run.stderr[-3000:]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["parallelism"] == "leaf-cell shards x2"
    assert line["launcher"]["self_launched"] and [a["ok"] for a in line["launcher"]["attempts"]] == [True]
    assert "kernel_ms" in line["roofline"] and "ms_per_batch" in line["export_sharded"]
    assert line["export_sharded"]["data_path_collectives_per_batch"] == 0


@pytest.mark.gpu
def test_bench_watchdog_retries_a_wedged_bootstrap_on_gloo_and_reports_a_dead_run():
    """(1) rank 1 stops right before the communicator bootstrap on the RCCL attempt: rank 0 waits for it in the rendezvous, nothing
    inside the ranks can end that.  The parent sees the bootstrap overdue, kills both ranks, starts two FRESH ranks on gloo and
    relays their line.  (2) the same on every backend: exactly one JSON line with an `error` field, exit status 1."""
    run, lines = _self_launched_bench(dict(S3_BENCH_HANG="1:alive:rccl", S3_BENCH_BOOT_TIMEOUT_S="15"))
    assert run.returncode == 0, run.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0
    att = lines[0]["launcher"]["attempts"]
    assert [a["backend"] for a in att] == ["rccl", "gloo"] and [a["ok"] for a in att] == [False, True]
    assert "bootstrap not finished" in att[0]["failure"]            # (rank 0 waits for rank 1 in the rendezvous: neither has a communicator)
    assert lines[0]["config"]["collectives"] == "gloo"
    run, lines = _self_launched_bench(dict(S3_BENCH_HANG="1:alive:any", S3_BENCH_BOOT_TIMEOUT_S="10"))
    assert run.returncode == 1
    assert len(lines) == 1 and lines[0]["value"] is None and "bootstrap not finished" in lines[0]["error"]
    assert [a["ok"] for a in lines[0]["launcher"]["attempts"]] == [False, False]


@pytest.mark.gpu
def test_bench_line_survives_a_failing_leg():
    """the legs beside the headline (batch records, transport, file) must not cost the driver its JSON line: a leg that raises is
    recorded as {"error": ...} in its place, the headline fields are those of a clean run"""
    import json
    env = dict(os.environ, S3_BENCH_FAIL_LEG="roofline_batches.T25")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cylinder3D_small", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1 and "kernel_ms" in line["roofline"]["pitched_copy"]
    assert "injected" in line["roofline_batches"]["T25"]["error"]
    assert "kernel_ms" in line["roofline_batches"]["T100"], "the other legs still run"
    assert "failure injected" in run.stderr


@pytest.mark.gpu
def test_sharded_export_matches_single_rank(tmp_path):
    """``ExportData`` with three ranks (leaf-cell shards: every rank interpolates its compact, cost-balanced share of the
    cells and of the vertices from the source rows that share references, one all-gather, rank 0 writes) produces the
    HDF5 file of the single-rank run bit for bit -- every output value is computed by the same kernel arithmetic whichever
    rank owns its cell.  Three processes share the one GPU, gloo carries the exchange."""
    from sparsespatialsampling_amd import h5io
    if h5io.native_lib() is None:
        pytest.importorskip("h5py")
    worker = os.path.join(ROOT, "tests", "sharded_export_worker.py")
    d1, d3 = str(tmp_path / "one"), str(tmp_path / "three")
    os.makedirs(d1), os.makedirs(d3)
    one = subprocess.run([sys.executable, worker, d1], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0 and "worker ok" in one.stdout, (one.stdout + one.stderr)[-3000:]
    env = dict(os.environ, S3_DIST_BACKEND="gloo")
    three = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                            "127.0.0.1", "--master-port", "29613", worker, d3], cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=900)
    assert three.returncode == 0 and three.stdout.count("worker ok") == 3, (three.stdout + three.stderr)[-3000:]
    assert sorted(os.listdir(d3)) == sorted(os.listdir(d1)), "one set of files, written by rank 0"
    with h5io.open_h5(os.path.join(d1, "case.h5"), "r") as a, h5io.open_h5(os.path.join(d3, "case.h5"), "r") as b:
        assert a.keys("data") == b.keys("data") and len(a.keys("data")) == 11
        for group in ("grid", "constant"):
            assert a.keys(group) == b.keys(group)
            for name in a.keys(group):
                assert np.array_equal(a.read(f"{group}/{name}"), b.read(f"{group}/{name}")), f"{group}/{name}"
        for t in a.keys("data"):
            assert a.keys(f"data/{t}") == b.keys(f"data/{t}") == ["U_center", "U_vertices", "p_center", "p_vertices"]
            for name in a.keys(f"data/{t}"):
                assert np.array_equal(a.read(f"data/{t}/{name}"), b.read(f"data/{t}/{name}")), f"data/{t}/{name}"
    assert open(os.path.join(d1, "case.xdmf")).read() == open(os.path.join(d3, "case.xdmf")).read()


@pytest.mark.gpu
def test_ranks_agree_on_a_gloo_group_when_rccl_refuses():
    """two ranks on ONE GPU ask for the RCCL communicator: RCCL refuses the duplicate device on both, the ranks find that out
    through the rendezvous store and carry the exchange steps of the refine over a gloo group instead; the bench line says
    so and the grid is the single-rank one"""
    import json
    env = dict(os.environ, S3_BENCH_SHARE_GPU="1")
    env.pop("S3_DIST_BACKEND", None)
    base = [os.path.join(ROOT, "bench.py"), "--workload", "cylinder3D_small", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29617"] + base + ["--gpus", "2"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    r2 = json.loads(two.stdout.strip().splitlines()[-1])
    assert r2["n_gpus"] == 2 and r2["config"]["collectives"].startswith("gloo (RCCL")
    assert r2["config"]["n_cells"] == 12942 and sum(r2["config"]["cells_per_rank"]) == 12942


@pytest.mark.gpu
@pytest.mark.parametrize("script,args,expect", [
    ("s3_for_synthetic_cylinder2D.py", [], ["metric_0.75.h5", "metric_0.75.xdmf", "metric_0.75_p_svd.h5"]),
    ("s3_for_synthetic_OAT15.py", ["250"], ["OAT15_synthetic_n_cells_25000.h5", "OAT15_synthetic_n_cells_25000.xdmf",
                                            "OAT15_synthetic_n_cells_25000_p_svd.h5"]),
])
def test_example_scripts_run_end_to_end(tmp_path, script, args, expect):
    """the example scripts -- the workflows of the reference's examples/s3_for_cylinder2D_Re100.py and s3_for_OAT15_airfoil.py on
    synthetic data: metric, grid generation, export of scalar and vector fields in batches, SVD -- run as a user runs them and leave
    their files (the OAT15 one with 250 instead of 2000 snapshots)"""
    run = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), str(tmp_path)] + args, cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    for name in expect:
        path = os.path.join(str(tmp_path), name)
        assert os.path.exists(path) and os.path.getsize(path) > 1000, (name, os.listdir(str(tmp_path)))
    from sparsespatialsampling_amd.data import Dataloader
    loader = Dataloader(str(tmp_path), expect[0])
    assert len(loader.write_times) == (250 if args else 400) and loader.vertices.shape[1] == 2


@pytest.mark.gpu
def test_process_exits_cleanly_after_transfers():
    """a process that has used the transfer lanes (staged upload of selected rows, staged download), a plan and the KNN index ends with
    status 0: the lanes are joined by s3_shutdown() from the bindings' atexit hook, before the interpreter and the HIP runtime go
    down (round 4 detached them after a core dump at exit; round 5 joins them in an orderly way).  Three fresh processes."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch as pt\n"
        "from sparsespatialsampling_amd import hipops, _lib\n"
        "rng = np.random.default_rng(0)\n"
        "x, c = rng.random((60000, 3)), rng.random((4000, 3))\n"
        "knn = hipops.KnnIndex(x)\n"
        "idx, dist = knn.query(c, 26)\n"
        "w = hipops.idw_weights(dist)\n"
        "host = pt.from_numpy(rng.standard_normal((60000, 200)).astype(np.float32))\n"
        "dev = hipops.to_device(host)\n"
        "plan = hipops.InterpPlan(idx, 60000, c); plan.set_weights(w)\n"
        "out = plan.interp(w, hipops.gather_rows(dev, pt.arange(60000, dtype=pt.int32, device='cuda'), hipops.padded_rows(60000, 200, pt.float32, 'cuda')))\n"
        "back = hipops.to_host(out)\n"
        "assert back.shape == (4000, 200)\n"
        "print('lanes joined at exit:', _lib.hip_lib().s3_shutdown() >= 0, flush=True)\n"
        "hipops.to_host(out)\n"                       # the pool is usable again after a shutdown; the atexit hook joins these lanes
        "print('done', flush=True)\n") % ROOT
    for _ in range(3):
        run = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert run.returncode == 0 and "done" in run.stdout and "Fatal" not in run.stderr, (run.returncode, run.stdout[-500:], run.stderr[-3000:])
