#!/usr/bin/env python3
"""
bench.py -- S^3 hot path on MI355X: snapshot interpolation throughput (+ refine wall-clock) on the synthetic
cylinder3D_Re3900 workload of BASELINE.json / SURVEY.md 8(d).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A *step* is one pass of the interpolation hot path over one batch of synthetic snapshots already resident in HBM IN THE FORM
THE API RECEIVES IT (interpolate_data(w, idx, data), reference export.py:446-468): a dense tensor [N_points, T_batch] fp32 with
a row for every point of the CFD mesh, read where it lies through ExportData's cached neighbour table
(s3_interp_planned_src) -> out [N_cells, T_batch] f64 in HBM.  The same launch on a pre-compacted, pitched copy of the
referenced rows (the layout ExportData uploads HOST batches into; the headline of rounds 1-3) is the sub-record
`roofline.pitched_copy`.
The grid comes from `SamplingTree.refine()` run on the GPU(s) before the timed region; its wall-clock is reported as
`refine_wall_s` in the same JSON line (second of two runs; the first one of a process, which also pays for the device
allocations, is `refine_first_run_wall_s`).

Multi-GPU (one process per GPU; the collectives run inside libs3hip.so on RCCL).  Under `torch.distributed.run` (WORLD_SIZE ==
--gpus) every rank runs main(); WITHOUT a launcher `python bench.py --gpus N` becomes the parent of N fresh ranks itself
(launch_ranks: the parent never touches the GPU, watches the ranks, kills a wedged attempt, retries once on gloo, and prints an
"error" JSON line + exit status 1 if that fails too):
* refine: every rank evaluates the KNN metric / gain of its 1/N slice of each batch of new cells, ONE grouped all-gather
  per batch -- the only exchange of a refinement step: the captured metric is reduced from the replicated arrays on every rank
  in a fixed block order (bit-identical for any N);
* interpolation (default `--shard cells`): the generated leaf cells are split into N spatially compact shards of equal
  cost (runs of the Hilbert-ordered tile plan, balanced by the bytes a shard moves per snapshot), every rank holds the
  KNN cache / plan of its shard and only the source rows that shard references, and interpolates the same snapshot
  batch -- no data-path collective; total work is fixed ("scaling": "strong").  `--shard snapshots` gives every rank
  all cells and its own snapshot batches instead ("weak").

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP events on the launch stream: mean, min, median, max
per launch, and the identity of the device they were taken on) and, at N = 1:
* `roofline_batches`: the same kernel family at the batch lengths the reference actually exports with -- 25 snapshots of a
  scalar (100-byte ragged rows) and of a 3-component field (300-byte rows), examples/s3_for_cylinder3D_Re3900.py:28-69 ->
  utils.py:204-226, and 100 snapshots -- each with its own algorithmic bytes;
* `device_resident_input`: a CUDA tensor [N, 1, T] fp32 as the reference hands batches over (every row of the CFD mesh,
  dense) -> ExportData's upload + neighbour table -> [Nc, T] f64 on the device, and the whole `_fit_data` incl. the download;
* `cpu_baseline` (the CPU oracle on a slice of this very workload at the bench's batch length), `refine_cpu_baseline`
  (`refine()` of THIS workload at full size with the oracle's kernels) and `end_to_end` (host tensors in -> host tensors
  out through ExportData, PCIe included; never `value`), with the three GPU / CPU ratios side by side in `gpu_over_cpu`.
"""
import argparse
import glob
import json
import os
import sys
import time
import types

# (robustness, no timed leg depends on it: torch's own copies to / from PAGEABLE host tensors of a MiB and more are served from the
# runtime's staging buffers instead of pinning the caller's pages on the fly -- that path produced rare GPU memory faults in round 5,
# HISTORY 9.  The library itself stages through its own page-locked buffers.  Set before the HIP runtime is even loaded.)
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4096")

import numpy as np  # noqa: E402
import torch as pt  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # the configuration BASELINE.json's metric is quoted on
    "cylinder3D_Re3900": dict(n=5_000_000, lo=[0.0, 0.0, 0.0], hi=[2.4, 2.0, 0.1 * np.pi], t_batch=1000,
                              uniform_levels=5, min_metric=0.75, seed=2),
    "cylinder3D_small": dict(n=300_000, lo=[0.0, 0.0, 0.0], hi=[2.4, 2.0, 0.1 * np.pi], t_batch=256,
                             uniform_levels=4, min_metric=0.6, seed=2),
    # HBM stress -- random centroids in the unit box, cell budget, batches of 16 snapshots (64-byte rows)
    "box5e7": dict(kind="box", n=50_000_000, t_batch=16, uniform_levels=5, n_cells_max=10_000_000, seed=3),
    "box5e7_small": dict(kind="box", n=5_000_000, t_batch=16, uniform_levels=4, n_cells_max=1_000_000, seed=3),
}


def synthetic_cylinder3d(cfg):
    """uniform random centroids in the cylinder3D box minus the cylinder; TKE-like metric with a wake"""
    rng = np.random.default_rng(cfg["seed"])
    lo, hi = np.asarray(cfg["lo"]), np.asarray(cfg["hi"])
    x = lo + rng.random((cfg["n"], 3)) * (hi - lo)
    dx, dy = x[:, 0] - 0.8, x[:, 1] - 1.0
    keep = dx * dx + dy * dy > 0.05 ** 2
    x, dx, dy = np.ascontiguousarray(x[keep]), dx[keep], dy[keep]
    r = np.sqrt(dx * dx + dy * dy)
    wake = np.exp(-(dy / 0.12) ** 2) * np.where(dx > 0, np.exp(-0.8 * dx), 0.0)
    metric = 0.02 + np.exp(-8.0 * r) + 0.9 * wake * (1 + 0.3 * np.sin(11.0 * dx)) * (1.0 + 0.1 * np.cos(40.0 * x[:, 2]))
    return x, metric


def synthetic_box(cfg):
    """uniform random centroids in the unit box, smooth metric peaking at the centre with short waves on top (C4)"""
    rng = np.random.default_rng(cfg["seed"])
    x = rng.random((cfg["n"], 3))
    r = np.sqrt(((x - 0.5) ** 2).sum(1))
    metric = 0.05 + np.exp(-6 * r) * (1 + 0.5 * np.sin(25 * x[:, 0]) * np.cos(17 * x[:, 1]))
    return x, metric


def build_case(name, cfg, geometry):
    """(points, metric, geometry objects, SamplingTree keyword arguments) of a workload"""
    if cfg.get("kind") == "box":
        x, metric = synthetic_box(cfg)
        geos = [geometry.CubeGeometry("domain", True, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0])]
        return x, metric, geos, dict(uniform_level=cfg["uniform_levels"], n_cells=cfg["n_cells_max"])
    x, metric = synthetic_cylinder3d(cfg)
    geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
            geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
    return x, metric, geos, dict(uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])


# ---- CPU baselines (rank 0, N = 1 only; the oracle is the checker of the tests, used here as the timed CPU port) --------
def host_cpu_counts():
    """what the host has, what this container may use, and (filled in by the caller) what the baseline ran on: a 256-thread host
    whose container is capped at 16 CPUs' worth of time is a 16-core baseline, whatever os.cpu_count() says"""
    quota = None
    try:                                                              # cgroup v2, then v1
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except (OSError, ValueError):
            pass
    return dict(host_cpus=os.cpu_count(), affinity_cpus=len(os.sched_getaffinity(0)), cgroup_cpus=quota)


def cpu_baseline(w, idx, data, k, used, seconds=10.0, n_sweep=100_000):
    """the CPU oracle (C + OpenMP restatement of export.py:446-468) on THIS workload: the bench's own neighbour table, the source
    rows it references, all snapshots of the bench's batch -- ALL cells, at the thread count a sweep over the first `n_sweep`
    cells found fastest (S3_BENCH_CPU_CELLS caps the cell count on hosts short of memory: the rows cost 4 * T bytes each)"""
    from oracle import s3_oracle as orc
    t = int(data.shape[1])

    from sparsespatialsampling_amd import hipops

    def slice_of(nc):
        # (downloads through the library's staged path: its own page-locked buffers, several host threads -- the 9.7 GB of the
        # referenced rows come down at the link's rate instead of a pageable copy's 7 GiB/s)
        i_s, w_s = hipops.to_host(idx[:nc]), hipops.to_host(w[:nc])
        rows, inv = np.unique(i_s, return_inverse=True)
        sub = hipops.to_host(data[used.long()[hipops.to_device(rows.astype(np.int64))]]).reshape(len(rows), 1, t)   # (`idx` holds positions in `used`)
        return w_s, inv.reshape(i_s.shape), sub, len(rows)

    nc_all = min(int(w.shape[0]), int(os.environ.get("S3_BENCH_CPU_CELLS", str(1 << 62))))
    w_s, inv, sub, _ = slice_of(min(nc_all, n_sweep))
    orc.interp(w_s, inv, sub)                  # warm
    # the baseline at ITS best thread count: under a container's CPU quota (16 CPUs' worth on the pool's 256-thread hosts) all
    # hardware threads are not the fastest choice -- 64 threads gave 700 M/s where 128 gave 450 and 16 gave 560
    all_threads, tried = orc.num_threads(), {}
    for n in sorted({all_threads, max(1, all_threads // 2), max(1, all_threads // 4), max(1, all_threads // 8)}, reverse=True):
        orc.set_num_threads(n)
        orc.interp(w_s, inv, sub)
        c0 = time.perf_counter()
        for _ in range(2):
            orc.interp(w_s, inv, sub)
        tried[n] = len(w_s) * t * 2 / (time.perf_counter() - c0) / 1e6
    best = max(tried, key=tried.get)
    orc.set_num_threads(best)
    del w_s, inv, sub
    w_s, inv, sub, n_rows = slice_of(nc_all)
    orc.interp(w_s, inv, sub)                  # warm (first touch of the output)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds or reps < 2:
        orc.interp(w_s, inv, sub)
        reps += 1
    dt = time.perf_counter() - t0
    orc.set_num_threads(all_threads)
    counts = host_cpu_counts()
    return dict(value=nc_all * t * reps / dt / 1e6, unit="Mcells*snapshots/s", cores=best, threads_used=best, kind="port", **counts,
                cells=nc_all, all_cells=bool(nc_all == int(w.shape[0])),
                threads_tried={str(n): round(v, 1) for n, v in tried.items()},
                sample=f"oracle/s3_oracle.c s3o_interp (OpenMP, {best} threads: the fastest of {sorted(tried)} on the first "
                       f"{min(nc_all, n_sweep)} cells) on the bench's own table: {nc_all} cells"
                       f"{' = all of them' if nc_all == int(w.shape[0]) else ''}, the {n_rows} source rows they reference, all {t} "
                       f"snapshots of the batch, {reps} passes, k={k}, fp32 in / f64 out; host: {counts['host_cpus']} hardware threads, "
                       f"cgroup quota {counts['cgroup_cpus']} CPUs")


def grid_sha(centers, levels, faces, nodes):
    """SHA-256 over the finished grid as refine() hands it over: cell centres, levels, face ids and node coordinates, each as
    contiguous little-endian bytes in the reference's dtypes (s_cube.py:734-772) -- two backends that print the same digest have
    built the same grid cell for cell, bit for bit"""
    import hashlib
    h = hashlib.sha256()
    for a in (centers, levels, faces, nodes):
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def refine_cpu_baseline(name, x, metric, geos, kw, n_leaf_gpu, sha_gpu):
    """`SamplingTree.refine()` of THIS workload at full size with the CPU oracle's kernels (tests/oracle_backend.py; the
    neighbour queries through the oracle's bucket grid, its stand-in for the reference's kd-tree: identical results to
    brute force, pinned by the same goldens) and the same host logic: the CPU port's grid-generation wall-clock"""
    import sparsespatialsampling_amd.s_cube as s_cube
    from oracle import s3_oracle as orc
    from tests.oracle_backend import OracleTreeBackend
    product = s_cube._make_backend
    s_cube._make_backend = lambda v, t, k: OracleTreeBackend(v, t, k, grid=True)
    all_threads, tried = orc.num_threads(), {}
    try:
        # (at the port's better thread count: all hardware threads are not the fastest choice under a container's CPU quota)
        for n in sorted({all_threads, max(1, all_threads // 4)}):
            orc.set_num_threads(n)
            t0 = time.perf_counter()
            tree = s_cube.SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
            tree.refine()
            tried[n] = time.perf_counter() - t0
            n_leaf = len(tree.all_centers)
            sha = grid_sha(tree.all_centers, tree.all_levels, tree.face_ids, tree.all_nodes)
            tree.close()
    finally:
        s_cube._make_backend = product
        orc.set_num_threads(all_threads)
    best = min(tried, key=tried.get)
    return dict(workload=f"{name} at full size: {len(x)} points -> {n_leaf} leaf cells", cpu_wall_s=tried[best],
                cores=best, threads_tried={str(n): round(v, 3) for n, v in tried.items()}, kind="port",
                knn="bucket grid (oracle/s3_oracle.c s3o_grid_*)", same_grid_size=bool(n_leaf == n_leaf_gpu),
                # full-size parity in the driver's own run: centres, levels, faces and nodes of the CPU port's grid hashed
                # against the HIP backend's (the reference itself cannot finish this size: BASELINE.md)
                grid_sha256_cpu_port=sha, grid_sha256_gpu=sha_gpu, same_grid_sha=bool(sha == sha_gpu))


def end_to_end(x, centers, k, batches=(25, 200)):
    """host tensors in -> host tensors out through ExportData._fit_data (KNN cache built once, then one batch per size;
    upload of the referenced rows, kernel, transpose, download): PCIe-inclusive rate, never `value`"""
    from sparsespatialsampling_amd.export import ExportData
    s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=None, levels=None,
                              metric=pt.zeros(len(x), dtype=pt.float64), size_initial_cell=1.0, save_path=".", save_name="bench",
                              grid_name="g")
    ex = ExportData(s, write_times=[str(i) for i in range(100000)], n_neighbors=k)
    coords = pt.from_numpy(x)
    out = {}
    for t in batches:
        data = pt.empty((len(x), 1, t), dtype=pt.float32).normal_()
        for _ in range(2):                                   # the first call builds the cache; two calls allocate BOTH pinned
            ex._fit_data(coords, data, "f", 10 ** 9)         # download buffers of this size (they are used alternately)
        pt.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 6
        for _ in range(reps):
            ex._fit_data(coords, data, "f", 10 ** 9)
        pt.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[f"T{t}"] = dict(ms_per_batch=dt * 1e3, Gcells_snapshots_per_s=len(centers) * t / dt / 1e9)
        del data
    out["note"] = ("pageable host tensor [N,1,T] fp32 in, host f64 tensor out; only the source rows the grid references are "
                   "uploaded; six batches back to back (steady state of an export: the download of a batch overlaps the upload "
                   "of the next one), all transfers finished before the clock stops")
    return out


def code_sha():
    """fingerprint of the native sources the kernels are built from (csrc/ + include/): the snapshot on the GPU box has no
    .git, so this is what ties a recorded profile to the code it was taken with"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "sparsespatialsampling_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def export_sharded(x, centers, k, comm, t=200, reps=3):
    """N > 1: what ExportData.export does with N ranks (SPMD), next to the bare step: every rank holds the dense device batch
    [N_points, 1, t] fp32, interpolates its shard of the cells from it and writes its rows -- transposed to snapshot-major --
    through its own PCIe link into the batch buffer all ranks map; timed up to the point where rank 0 could hand the buffer
    to the HDF5 writer (barrier-bracketed, max over ranks).  No value crosses xGMI."""
    from sparsespatialsampling_amd.export import ExportData
    s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=None, levels=None,
                              metric=pt.zeros(len(x), dtype=pt.float64), size_initial_cell=1.0, save_path=".", save_name="bench",
                              grid_name="g")
    ex = ExportData(s, write_times=[str(i) for i in range(100000)], n_neighbors=k)
    ex._interpolated_metric = True       # (the one-time metric of the original grid is not part of a batch: no gather to the root here)
    coords = pt.from_numpy(x)
    data = pt.empty((len(x), 1, t), dtype=pt.float32, device="cuda").normal_(generator=pt.Generator(device="cuda").manual_seed(99))
    for _ in range(2):                                       # cache, plans, both shared buffers
        ex._fit_data(coords, data, "f", 10 ** 9)
    pt.cuda.synchronize()
    comm.barrier()
    c0 = comm.n_collectives
    t0 = time.perf_counter()
    for _ in range(reps):
        ex._fit_data(coords, data, "f", 10 ** 9)
    pt.cuda.synchronize()
    comm.barrier()
    dt = comm.allreduce_max(time.perf_counter() - t0) / reps
    collectives = (comm.n_collectives - c0) / reps              # counted by the communicator: all-gathers + gathers to the root
    table = ex._table_centers
    return dict(t_batch=t, ms_per_batch=dt * 1e3, Gcells_snapshots_per_s=len(centers) * t / dt / 1e9,
                cells_on_this_rank=int(len(table.shard.mine)), data_path_collectives_per_batch=collectives,
                host_bytes_written_by_this_rank=int(len(table.shard.mine)) * t * 8,
                shared_host_buffer=ex._shared is not None,
                direct_device_writes=bool(ex._shared and all(b.device_ptr is not None for b in ex._shared.values())),
                note="dense device batch in -> the ranks' rows in ONE shared host buffer (snapshot-major, file order), "
                     "ready for the writer of rank 0; each rank over its own PCIe link")


# the batch lengths the reference exports with (examples/s3_for_cylinder3D_Re3900.py:28-69, utils.py:204-226)
BATCH_SHAPES = [("T25", 25, "25 snapshots of a scalar field: 100-byte ragged rows"),
                ("T25x3", 75, "25 snapshots of a 3-component field: 300-byte rows"),
                ("T100", 100, "100 snapshots of a scalar field: 400-byte rows")]
CHILD_LAUNCHES = 3          # launches per group of a PMC child pass (after one warm launch of the headline)


def recorded_traffic(workload_key):
    """HBM bytes per launch of the dominant kernel as RECORDED in the committed PMC passes (profiles/rNN/summary.json:
    FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE from separate rocprofv3 --pmc runs of this
    command); -> (bytes, file, stale): `stale` when the native sources changed since that collection (`code_sha`);
    (None, None, None) when no pass was recorded for this workload"""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*", "summary.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except (OSError, ValueError):
            continue
        if rec.get("workload_key") == workload_key and "traffic_bytes_per_launch" in rec:
            return rec["traffic_bytes_per_launch"], os.path.relpath(f, ROOT), rec.get("code_sha") != code_sha()
    return None, None, None


def measure_traffic(argv, kernel_prefix):
    """HBM bytes per launch of the headline kernel, measured IN THIS RUN: before this process touches the GPU it starts two child
    processes of itself under `rocprofv3 --pmc <counter> --kernel-trace` (one pass per counter: FETCH_SIZE and WRITE_SIZE do not fit
    one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), each a short run of the same workload (3 launches), and averages the
    counter over the launches of the headline kernel.  traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-byte
    requests at 64 bytes; both in KiB).  -> dict or None (no rocprofv3, a failing pass, a profiler already attached)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    out, groups, t0 = {}, {}, time.perf_counter()
    group_names = [f"{name}/{where}" for name, _, _ in BATCH_SHAPES for where in ("inplace", "pitched")]
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="s3_bench_pmc_")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__)] + argv + ["--traffic-child"]
            run = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL,
                                 stderr=subprocess.PIPE, text=True, timeout=float(os.environ.get("S3_BENCH_PMC_TIMEOUT_S", "300")))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if run.returncode != 0 or not files:
                print(f"[bench] PMC pass {counter} failed (rc {run.returncode}): {run.stderr[-300:]}", file=sys.stderr)
                return None
            rows = sorted(((int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(files[0]))
                           if r["Counter_Name"] == counter and kernel_prefix in r["Kernel_Name"] and "permute" not in r["Kernel_Name"]))
            vals = [v for _, v in rows]
            if not vals:
                print(f"[bench] PMC pass {counter}: no launch of {kernel_prefix}", file=sys.stderr)
                return None
            # the child launches in a fixed order: the headline (1 warm + CHILD_LAUNCHES), then per batch shape CHILD_LAUNCHES in place
            # and CHILD_LAUNCHES on the pitched copy
            head = vals[:1 + CHILD_LAUNCHES][1:]
            out[counter] = (sum(head) / len(head), len(head))
            rest = vals[1 + CHILD_LAUNCHES:]
            if len(rest) == len(group_names) * CHILD_LAUNCHES:
                for gi, gname in enumerate(group_names):
                    g = rest[gi * CHILD_LAUNCHES:(gi + 1) * CHILD_LAUNCHES]
                    groups.setdefault(gname, {})[counter] = sum(g) / len(g)
            elif rest:
                print(f"[bench] PMC pass {counter}: {len(rest)} launches behind the headline's, expected {len(group_names) * CHILD_LAUNCHES}: "
                      f"batch shapes not attributed", file=sys.stderr)
        except (OSError, subprocess.SubprocessError, ValueError, KeyError) as err:
            print(f"[bench] PMC pass {counter} failed: {type(err).__name__}: {err}", file=sys.stderr)
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = out["FETCH_SIZE"][0], out["WRITE_SIZE"][0]
    batches = {g: dict(traffic=(2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, FETCH_SIZE_KiB_per_launch=c["FETCH_SIZE"],
                       WRITE_SIZE_KiB_per_launch=c["WRITE_SIZE"]) for g, c in groups.items() if len(c) == 2}
    return dict(traffic=(2.0 * fetch + write) * 1024.0, FETCH_SIZE_KiB_per_launch=fetch, WRITE_SIZE_KiB_per_launch=write,
                batches=batches,
                launches_averaged=[out["FETCH_SIZE"][1], out["WRITE_SIZE"][1]], seconds=time.perf_counter() - t0,
                how="measured in this run: two child processes of this command under rocprofv3 --pmc (one pass per counter, kernel-trace "
                    "only), 2 x FETCH_SIZE + WRITE_SIZE")


def copy_bandwidth_gbs(n_bytes=2 << 30, reps=5):
    """device-to-device copy rate of this box (read + write bytes per second), the practical HBM ceiling next to the
    8 TB/s datasheet peak (report both fractions)"""
    src = pt.empty(n_bytes // 4, dtype=pt.float32, device="cuda").normal_()
    dst = pt.empty_like(src)
    dst.copy_(src)
    pt.cuda.synchronize()
    a, b = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    pt.cuda.synchronize()
    return 2.0 * n_bytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def yardsticks(hipops, plan, data, steps, warmup):
    """what this chip's memory system gives hand-written kernels of this library (the runtime's blit is a soft
    yardstick): a float4 copy, a 7-read / 2-write mix (the headline launch's own 78 % / 22 %), reads only, and the LOADS of the
    headline's tile plan on the headline's table with nothing else in the kernel (s3_yard_stream, s3_yard_plan_loads) -- the last
    one in the shift kernel's schedule (one 128-byte line of a row per visit) and with two consecutive lines per visit."""
    res = {}
    n_bytes = 4 << 30
    src = pt.empty(n_bytes // 4, dtype=pt.float32, device="cuda").normal_()
    dst = pt.empty(n_bytes // 4, dtype=pt.float32, device="cuda")
    for key, r, w in (("copy_float4", 1, 1), ("mix_7r2w", 7, 2), ("read_only", 4, 0)):
        moved = hipops.yard_stream(src, dst, r, w)
        ms = launch_times_ms(lambda: hipops.yard_stream(src, dst, r, w), max(steps // 2, 5), 2)
        res[key] = dict(GBs=sum(moved) / (float(np.mean(ms)) * 1e-3) / 1e9, bytes_read=moved[0], bytes_written=moved[1], **ms_stats(ms))
    del src, dst
    # (the load probe models the chunk kernels: rows of at least one 128-byte line on a 16-byte aligned pitch, 64-cell tiles)
    row_bytes = int(data.shape[1]) * data.element_size()
    if plan is not None and row_bytes >= 128 and row_bytes % 16 == 0:
        for key, variant in (("plan_loads_one_line_per_visit", 0), ("plan_loads_two_lines_per_visit", 1)):
            staged = plan.yard_loads(data, variant)
            ms = launch_times_ms(lambda: plan.yard_loads(data, variant), steps, warmup)
            res[key] = dict(staged_bytes=staged, staged_GBs=staged / (float(np.mean(ms)) * 1e-3) / 1e9, **ms_stats(ms))
    res["note"] = ("GBs = bytes read + written per second of a hand-written streaming kernel over 4-GiB buffers (short-lived workgroups, one "
                   "contiguous block each, nontemporal loads / stores: the fastest form measured, tools/copy_probe.hip); plan_loads: the headline's tile plan on the headline's table, loads only (256 threads, two workgroups per CU, "
                   "two rolling sets of sixteen 16-byte vectors per lane), staged bytes include the halo the L2 serves")
    return res


_SMI = {}


def query_rocm_smi():
    """serial / unique id / clock levels of card 0 from rocm-smi -- called BEFORE this process touches the GPU (rocm-smi is a
    Python script: starting it means an exec in a forked child, which the GPU pool refuses once the parent has initialised the
    device), and not at all under a profiler whose preloaded library has initialised it already"""
    if _SMI or "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        _SMI.setdefault("rocm_smi", "not queried (profiler attached)")
        return
    try:
        import subprocess
        out = subprocess.run(["rocm-smi", "--showuniqueid", "--showserial", "--showclocks", "--json"], capture_output=True,
                             text=True, timeout=20).stdout
        card = json.loads(out).get("card0", {})
        _SMI["unique_id"] = card.get("Unique ID")
        _SMI["serial"] = card.get("Serial Number")
        _SMI["clocks"] = {k.split(" ")[0]: v for k, v in card.items() if "clock" in k.lower() and "level" in k.lower()}
    except Exception as err:                      # no rocm-smi on the box / no permission: the torch fields remain
        _SMI["rocm_smi"] = f"unavailable: {type(err).__name__}"


def device_identity():
    """which card the numbers were taken on: launch times of one binary differ by several per cent between the boxes of a
    pool, so a timing without the device behind it cannot be matched to a profile"""
    prop = pt.cuda.get_device_properties(pt.cuda.current_device())
    ident = dict(name=prop.name, gcn_arch=getattr(prop, "gcnArchName", None), compute_units=prop.multi_processor_count,
                 hbm_gib=round(prop.total_memory / 2 ** 30, 1), uuid=str(getattr(prop, "uuid", "")) or None,
                 hostname=os.uname().nodename)
    ident.update(_SMI)
    return ident


def launch_times_ms(fn, steps, warmup):
    """HIP-event time of every one of `steps` launches of fn() on the current stream (after `warmup` untimed ones)"""
    for _ in range(warmup):
        fn()
    pt.cuda.synchronize()
    ev = [(pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    pt.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def ms_stats(ms):
    return dict(kernel_ms=float(np.mean(ms)), kernel_ms_min=float(np.min(ms)), kernel_ms_median=float(np.median(ms)),
                kernel_ms_max=float(np.max(ms)), launches=len(ms))


def planned_kernel_name(t_elems, k, plan_tiles, pitch_elems=None, table_bytes=0):
    """the kernel s3_interp_planned / s3_interp_planned_src dispatches a batch of fp32 rows of t_elems elements with a row pitch
    of pitch_elems elements to (csrc/interp_plan.hip: planned_dispatch / launch_planned); None: whole 128-byte lines"""
    pitch = pitch_elems if pitch_elems is not None else (t_elems + 31) // 32 * 32
    even = "true" if t_elems % 2 == 0 else "false"
    if pitch % 4:                                                    # rows on element boundaries only: the persistent kernel
        return f"interp_planned_stream_kernel<float,{k},false,{even},false>"
    vecs = (t_elems + 3) // 4
    if (2 <= vecs <= 4 and k in (8, 26) and plan_tiles >= int(os.environ.get("S3_STREAM_MIN_TILES", "64"))
            and os.environ.get("S3_SHORT_STREAM", "1" if table_bytes <= 1 << 30 else "0") == "1"):
        return f"interp_planned_stream_kernel<float,{k},true,{even},true>"       # the persistent kernel's narrow layout
    if vecs <= 4:
        return ("interp_planned_short_quad_kernel<float,7>" if vecs == 4 and not os.environ.get("S3_SHORT_NO_QUAD")
                else "interp_planned_short_reg_kernel<float,26>")
    chunks = (t_elems + 31) // 32
    shift = pitch % 32 != 0 and os.environ.get("S3_INPLACE_SHIFT", "1") != "0"     # rows off the 128-byte grid: whole lines, phase undone in LDS
    if (chunks <= int(os.environ.get("S3_STREAM_MAX_CHUNKS", "24")) and k in (8, 26) and plan_tiles >= int(os.environ.get("S3_STREAM_MIN_TILES", "64"))
            and not (shift and chunks >= int(os.environ.get("S3_SHIFT_MIN_CHUNKS", "6")))):
        return f"interp_planned_stream_kernel<float,{k},true,{even},false>"
    if shift:
        return "interp_planned_shift_kernel<float>"
    return "interp_planned_kernel<float,64>"


def batch_record(hipops, plan, w, used, n_points, nc, k, row_len, label, workload_key, steps, warmup, gen, measured=None):
    """roofline sub-record of one batch shape: a dense [n_points, row_len] fp32 batch read in place (the API's form); the same
    launch on the pitched copy of the referenced rows (the layout ExportData uploads host batches into) beside it"""
    n_rows = int(used.numel())
    table = pt.empty((n_points, row_len), dtype=pt.float32, device="cuda")
    table.normal_(generator=gen)
    out = pt.empty((nc, row_len), dtype=pt.float64, device="cuda")
    ms = launch_times_ms(lambda: plan.interp_src(table, out=out), steps, warmup)
    b_alg = n_rows * row_len * 4 + nc * row_len * 8 + nc * k * (4 + 8)
    st = ms_stats(ms)
    # HBM bytes per launch: measured in THIS run (the PMC child passes launched these shapes behind the headline) or absent -- a figure
    # recorded by an earlier collection is no longer printed (`traffic_stale` must not be true in the line)
    shape = workload_key.rsplit("/", 1)[-1]
    m_in = (measured or {}).get("batches", {}).get(f"{shape}/inplace")
    m_p = (measured or {}).get("batches", {}).get(f"{shape}/pitched")
    traffic = m_in["traffic"] if m_in else None
    src, stale = ("measured in this run (PMC child passes, 2 x FETCH_SIZE + WRITE_SIZE)" if m_in else None), (False if m_in else None)
    rec = dict(rows=label + ", dense [N, L] batch of all points read in place", row_bytes=row_len * 4, pitch_bytes=row_len * 4,
               kernel=planned_kernel_name(row_len, k, plan.n_tiles, row_len, table_bytes=n_points * row_len * 4),
               algorithmic_bytes=b_alg, achieved=b_alg / (st["kernel_ms"] * 1e-3) / 1e9, unit="GB/s",
               frac=b_alg / (st["kernel_ms"] * 1e-3) / 8e12, frac_best_launch=b_alg / (st["kernel_ms_min"] * 1e-3) / 8e12,
               Gcells_snapshots_per_s=nc * row_len / (st["kernel_ms"] * 1e-3) / 1e9, traffic=traffic,
               traffic_source=src, traffic_stale=stale, traffic_over_algorithmic=None if traffic is None else traffic / b_alg, **st)
    data = hipops.gather_rows(table, used, hipops.padded_rows(n_rows, row_len, pt.float32, "cuda"))
    del table
    pms = ms_stats(launch_times_ms(lambda: plan.interp(w, data, out=out), steps, warmup))
    rec["pitched_copy"] = dict(pitch_bytes=int(data.stride(0)) * 4, kernel=planned_kernel_name(row_len, k, plan.n_tiles),
                               frac=b_alg / (pms["kernel_ms"] * 1e-3) / 8e12, traffic=m_p["traffic"] if m_p else None,
                               traffic_over_algorithmic=m_p["traffic"] / b_alg if m_p else None, **pms)
    del data, out
    return rec


def optional_leg(name, fn):
    """a leg beside the headline: its failure (a full /tmp, a host without free pinned memory ...) must not cost the bench line
    -- the error is recorded in the leg's place and printed to stderr"""
    try:
        if os.environ.get("S3_BENCH_FAIL_LEG") == name:        # (test hook: tests/test_gpu_refine.py)
            raise RuntimeError("failure injected by S3_BENCH_FAIL_LEG")
        return fn()
    except Exception as err:                                   # noqa: BLE001 -- anything: the headline is already measured
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {"error": f"{name}: {type(err).__name__}: {err}"}


def export_to_file(x, metric, tree_out, k, t=25, n_batches=8):
    """the whole product path with the file at its end: ExportData.export() of `n_batches` host batches of `t` snapshots of a
    scalar field (the reference's loop, examples/s3_for_cylinder3D_Re3900.py:28-69 -> utils.py:204-226) into a real HDF5 +
    XDMF pair in a temporary directory -- KNN cache, upload, interpolation, transpose, download, background writer, close.
    `tree_out`: (centers, vertices, faces, levels, width) of the generated grid."""
    import shutil
    import tempfile
    from sparsespatialsampling_amd.export import ExportData
    centers, vertices, faces, levels, width = tree_out
    d = tempfile.mkdtemp(prefix="s3_bench_")
    try:
        s = types.SimpleNamespace(n_dimensions=3, faces=faces, centers=centers, vertices=vertices, levels=levels,
                                  metric=pt.from_numpy(metric), size_initial_cell=width, save_path=d, save_name="bench", grid_name="g")
        ex = ExportData(s, write_times=[str(i) for i in range(t * n_batches)], n_neighbors=k)
        coords = pt.from_numpy(x)
        batches = [pt.empty((len(x), 1, t), dtype=pt.float32).normal_() for _ in range(2)]
        t0 = time.perf_counter()
        per = []
        for b in range(n_batches):
            tb = time.perf_counter()
            ex.export(coords, batches[b % 2], "p", n_snapshots_total=t * n_batches)
            per.append((time.perf_counter() - tb) * 1e3)
        total = time.perf_counter() - t0                       # the last export() closes the file: every value is on disk
        size = os.path.getsize(os.path.join(d, "bench.h5"))
        steady = float(np.median(per[2:-1])) if n_batches > 4 else float(np.median(per))
        return dict(t_batch=t, batches=n_batches, total_s=total, Gcells_snapshots_per_s=len(centers) * t * n_batches / total / 1e9,
                    ms_per_export_call=[round(p, 2) for p in per], ms_per_export_call_steady=steady,
                    Gcells_snapshots_per_s_steady=len(centers) * t / (steady * 1e-3) / 1e9,
                    file_MB=size / 1e6, file_MB_per_s=size / total / 1e6, directory=os.path.dirname(d),
                    note="first call builds the KNN cache and writes the grid, the last one waits for the writer and closes the file; "
                         "steady = median of the calls in between")
    finally:
        shutil.rmtree(d, ignore_errors=True)


def numbering_follows_space(hipops, x, idx, used, w, centers, k, row_len, steps, warmup, gen):
    """the same in-place launch on a mesh whose point NUMBERING follows space (the points renumbered along their Hilbert
    curve -- what a block-structured or bandwidth-reduced CFD mesh looks like; the bench cloud is numbered at random, the
    worst case for short dense rows: a 100-byte row starts anywhere in a 128-byte line and its neighbours in memory belong to
    cells far away).  Same grid, same neighbour sets, same values per cell; only where the rows lie in the table changes."""
    n = len(x)
    order = hipops.spatial_order(x).long()                           # position -> point
    new_id = pt.empty(n, dtype=pt.int32, device="cuda")
    new_id[order] = pt.arange(n, dtype=pt.int32, device="cuda")
    idx_new = new_id[used.long()[idx.long()]].contiguous()             # neighbour table in the new numbering (full table ids)
    used2, remap2 = hipops.referenced_rows([idx_new], n, coords=None)
    hipops.remap_indices(idx_new, remap2)
    plan = hipops.InterpPlan(idx_new, int(used2.numel()), centers)
    plan.set_weights(w)
    plan.set_source_ids(used2.contiguous(), n)
    table = pt.empty((n, row_len), dtype=pt.float32, device="cuda")
    table.normal_(generator=gen)
    out = pt.empty((len(centers), row_len), dtype=pt.float64, device="cuda")
    st = ms_stats(launch_times_ms(lambda: plan.interp_src(table, out=out), steps, warmup))
    b_alg = int(used2.numel()) * row_len * 4 + len(centers) * row_len * 8 + len(centers) * k * 12
    plan.close()
    return dict(frac=b_alg / (st["kernel_ms"] * 1e-3) / 8e12, note="points renumbered along their Hilbert curve", **st)


def device_resident_input(x, centers, k, t_list, bare_ms):
    """a CUDA tensor [N, 1, T] fp32 as the reference hands batches over (export.py:128-167: every row of the CFD mesh, dense)
    -> ExportData: `interp_ms` = upload step + neighbour table on the device (HIP events; [Nc, T] f64 stays in HBM),
    `fit_data_ms` = the whole _fit_data incl. transpose and download to the host.  `bare_kernel_ms`: the headline's launch
    (s3_interp_planned_src on the same kind of tensor, without ExportData around it)."""
    from sparsespatialsampling_amd.export import ExportData, _as_float
    s = types.SimpleNamespace(n_dimensions=3, faces=None, centers=pt.from_numpy(centers), vertices=None, levels=None,
                              metric=pt.zeros(len(x), dtype=pt.float64), size_initial_cell=1.0, save_path=".", save_name="bench",
                              grid_name="g")
    ex = ExportData(s, write_times=[str(i) for i in range(100000)], n_neighbors=k)
    coords = pt.from_numpy(x)
    out = {}
    for t in t_list:
        data = pt.empty((len(x), 1, t), dtype=pt.float32, device="cuda").normal_()
        for _ in range(2):                                   # builds the cache on the first call; two calls allocate both
            ex._fit_data(coords, data, "f", 10 ** 9)         # pinned download buffers of this size (used alternately)

        def on_device():
            batch, in_place = ex._upload(_as_float(data))
            return ex._table_centers.apply(batch, True, full_table=in_place)
        ms = launch_times_ms(on_device, 10, 2)
        pt.cuda.synchronize()
        t0 = time.perf_counter()
        calls = []
        for _ in range(3):
            c0 = time.perf_counter()
            ex._fit_data(coords, data, "f", 10 ** 9)
            calls.append((time.perf_counter() - c0) * 1e3)   # (host time of the call: the download may still be running)
        pt.cuda.synchronize()
        fit_ms = (time.perf_counter() - t0) / 3 * 1e3
        rec = dict(interp_ms=float(np.mean(ms)), interp_ms_min=float(np.min(ms)), interp_ms_median=float(np.median(ms)),
                   fit_data_ms=fit_ms, Gcells_snapshots_per_s=len(centers) * t / (float(np.mean(ms)) * 1e-3) / 1e9,
                   rows_read_in_place=bool(ex._upload(_as_float(data))[0].data_ptr() == data.data_ptr()),
                   download_buffers_pinned=bool(all(b.is_pinned() for b in ex._host_stage.values())),
                   fit_data_calls_ms=[round(c, 1) for c in calls])
        if t in bare_ms:
            rec["bare_kernel_ms"] = bare_ms[t]
            rec["over_bare_kernel"] = rec["interp_ms"] / bare_ms[t]
        out[f"T{t}"] = rec
        del data
    out["note"] = ("CUDA tensor [N, 1, T] fp32 (dense: all rows of the CFD mesh, row pitch = T elements) in; interp_ms: "
                   "ExportData._upload + neighbour table, output [Nc, T] f64 in HBM; fit_data_ms adds transpose + pinned download")
    return out


def bench_svd(args, json_fd):
    """`--workload svd`: the weighted Gram matrix of the interpolated snapshot matrix on the f64 matrix cores
    (s3_weighted_gram: the kernel of utils.compute_svd, reference utils.py:302-346) at the bench grid's size -- 461 130 cells x
    1000 snapshots f64 resident in HBM, a step = one launch.  MFMA-bound: algorithmic flops N * T * (T + 1) (upper triangle)
    against AMD's datasheet peak for f64 matrix operations (78.6 TFLOP/s; the guide lists none) and against the rate a bare
    loop of the same instruction sustains on this pool (46 TFLOP/s, tools/mfma_f64_peak.hip).  Eigen-solve and mode GEMM of
    compute_svd: the eigen-solve of the T x T matrix is a vendor-library call (rocSOLVER's dsyevd behind s3_sym_eig), the mode GEMM is
    s3_centered_gemm on the same matrix cores (`mode_gemm`); the whole call is timed in `compute_svd_s`, not in `value`."""
    from sparsespatialsampling_amd import svd, metrics
    n, t = 461_130, args.t_batch or 1000
    gen = pt.Generator(device="cuda").manual_seed(7)
    x = pt.empty((n, t), dtype=pt.float64, device="cuda").normal_(generator=gen)
    x += pt.linspace(0, 3, t, dtype=pt.float64, device="cuda").sin() * pt.empty((n, 1), dtype=pt.float64, device="cuda").normal_(generator=gen)
    w = pt.empty(n, dtype=pt.float64, device="cuda").uniform_(0.05, 0.55, generator=gen)
    mean = metrics.temporal_mean(x)
    ms = launch_times_ms(lambda: svd.weighted_gram(x, mean, w), args.steps, args.warmup)
    pt.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        svd.weighted_gram(x, mean, w)
    pt.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    flops = float(n) * t * (t + 1)
    st = ms_stats(ms)
    t1 = time.perf_counter()
    s_, u_, v_ = svd.compute_svd(x, w, rank=50)
    pt.cuda.synchronize()
    svd_s = time.perf_counter() - t1                       # (first call of the process: the solver library initialises)
    steady, parts = [], {}
    for _ in range(3):
        t1 = time.perf_counter()
        svd.compute_svd(x, w, rank=50)
        pt.cuda.synchronize()
        steady.append(time.perf_counter() - t1)
    gram = svd.weighted_gram(x, mean, w)
    pt.cuda.synchronize()
    t1 = time.perf_counter()
    svd._eigh(gram)
    pt.cuda.synchronize()
    parts["eigen_solve_ms"] = (time.perf_counter() - t1) * 1e3
    # the ONE library call of the path, said so in the line itself: the symmetric eigen-solve of the T x T Gram matrix
    # is rocSOLVER's dsyevd, reached through the C ABI (s3_sym_eig looks it up with dlopen; no torch operator); everything else of
    # compute_svd is this repository's kernels
    from sparsespatialsampling_amd import _lib as _s3lib
    parts["eigh"] = "library"
    parts["eigh_library"] = ("rocSOLVER dsyevd behind s3_sym_eig (C ABI, dlopen)" if _s3lib.hip_lib().s3_sym_eig_available()
                             else "LAPACK on the host (torch.linalg.eigh): rocSOLVER not loadable")
    parts["eigh_share_of_compute_svd"] = parts["eigen_solve_ms"] * 1e-3 / float(np.median(steady))
    achieved = flops / (st["kernel_ms"] * 1e-3) / 1e12
    # the tall GEMMs of the same SVD on the same matrix cores (s3_centered_gemm): the mode GEMM U = (X - mean) V S^-1 at the rank
    # compute_svd was asked for (50 columns occupy half of a 128-column tile) and at full width (the shape of a deflation level)
    gemm = {}
    for r in (50, 128, t):
        b = pt.empty((t, r), dtype=pt.float64, device="cuda").normal_(generator=gen)
        gms = ms_stats(launch_times_ms(lambda: svd.centered_gemm(x, mean, b), max(3, args.steps // 4), 1))
        tf = 2.0 * n * t * r / (gms["kernel_ms"] * 1e-3) / 1e12
        gemm[f"r{r}"] = dict(columns=r, algorithmic_flops=2.0 * n * t * r, TFLOPs=tf, frac_of_sustained=tf / 46.0, frac_of_datasheet=tf / 78.6, **gms)
        del b
    res = {"metric": "TFLOP/s weighted Gram matrix (f64 matrix cores)", "value": flops * args.steps / elapsed / 1e12, "unit": "TFLOP/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"svd: weighted Gram matrix of {n} cells x {t} snapshots f64 (the bench grid's interpolated field)",
                      "n_cells": n, "t_batch": t},
           "device": device_identity(),
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": 78.6, "unit": "TFLOP/s", "frac": achieved / 78.6,
                        "peak_source": "AMD MI355X datasheet, f64 matrix (the CDNA4 guide lists no f64 figure)",
                        "sustained_instruction_rate": 46.0, "frac_of_sustained": achieved / 46.0,
                        "sustained_source": "bare v_mfma_f64_16x16x4_f64 loop, operands in registers (tools/mfma_f64_peak.hip, HISTORY 5.6)",
                        "kernel": "gram_block_kernel", "algorithmic_flops": flops, "traffic": None, **st},
           "mode_gemm": dict(kernel="centered_gemm_kernel", shapes=gemm,
                             note="C[N, r] = (X - mean 1^T) B, X [N, T] f64 as the interpolation left it, B [T, r]; flops 2 N T r"),
           "eigh": "library", "compute_svd_s": svd_s, "compute_svd_steady_s": float(np.median(steady)), "compute_svd_parts": parts, "compute_svd_rank": int(len(s_)),
           "compute_svd_note": "mean + Gram kernel + eigen-solve (rocSOLVER, the one library call) + mode GEMM (s3_centered_gemm), rank 50; "
                               "compute_svd_s: first call of the process (the solver library initialises), compute_svd_steady_s: median of the next three"}
    os.write(json_fd, (json.dumps(res) + "\n").encode())


# ---- N > 1 without a launcher: this process becomes the PARENT of N fresh ranks ---------------------------------------------
# `python bench.py --gpus N` with no WORLD_SIZE in the environment (the way the driver runs `--gpus 1`) must still print its one
# JSON line.  The parent parses the arguments and NEVER touches the GPU (it imports torch, which does not initialise HIP; every HIP
# call happens in the children): it starts N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, each in a process group of its
# own), relays rank 0's line, and is their WATCHDOG -- `ncclCommInitRank` has no timeout and nothing inside a rank can see that
# another rank is wedged.  Children report two milestones through files in a scratch directory ("alive": imports done and device
# selected; "boot": communicator created); if the bootstrap takes longer than S3_BENCH_BOOT_TIMEOUT_S (120) after all ranks were
# alive, if a rank exits with an error, or if the run exceeds S3_BENCH_RUN_TIMEOUT_S (1500), ALL children are killed (their process
# groups: SIGTERM, then SIGKILL) and N FRESH children are started ONCE with S3_DIST_BACKEND=gloo (the exchange steps through a
# gloo group; no process that has initialised the GPU is ever re-exec'ed).  If that attempt fails too the parent prints a JSON
# line with an "error" field and exits with status 1.
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _mark(stage):
    """child side of the watchdog protocol: milestone `stage` of this rank reached"""
    d = os.environ.get("S3_BENCH_WATCH_DIR")
    if d:
        open(os.path.join(d, f"{stage}_{os.environ.get('RANK', '0')}"), "w").close()
    # test hook, S3_BENCH_HANG=<rank|*>:<stage>:<backend|any>: that rank (every rank) stops (sleeps) when it reaches the stage on that backend
    hang = os.environ.get("S3_BENCH_HANG", "").split(":")
    if len(hang) == 3 and hang[0] in ("*", os.environ.get("RANK", "0")) and hang[1] == stage:
        backend = "gloo" if os.environ.get("S3_DIST_BACKEND") == "gloo" else "rccl"
        if hang[2] in ("any", backend):
            print(f"[bench] rank {os.environ.get('RANK', '0')}: S3_BENCH_HANG -- sleeping at stage '{stage}'", file=sys.stderr, flush=True)
            while True:
                time.sleep(3600)


def _kill_ranks(procs):
    import signal
    for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
        alive = [q for q in procs if q.poll() is None]
        if not alive:
            return
        for q in alive:
            try:
                os.killpg(q.pid, sig)                      # (start_new_session: the child leads a group of its own)
            except (ProcessLookupError, PermissionError):
                pass
        t_end = time.monotonic() + grace
        while time.monotonic() < t_end and any(q.poll() is None for q in procs):
            time.sleep(0.05)


def _run_ranks(n, argv, extra_env, boot_timeout, run_timeout, start_timeout):
    """one attempt: N fresh children -> (rank 0's JSON line or None, reason of the failure or None, seconds)"""
    import shutil
    import subprocess
    import tempfile
    watch = tempfile.mkdtemp(prefix="s3_bench_watch_")
    port = _free_port()
    procs, reason = [], None
    t0 = time.monotonic()
    try:
        out0 = open(os.path.join(watch, "rank0.stdout"), "wb")
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), S3_BENCH_WATCH_DIR=watch, S3_BENCH_CHILD="1", **extra_env)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdin=subprocess.DEVNULL,
                                          stdout=out0 if r == 0 else sys.stderr, start_new_session=True))
        out0.close()
        t_alive = None
        while True:
            codes = [q.poll() for q in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                reason = "rank %d exited with status %d" % bad[0]
                break
            if all(c == 0 for c in codes):
                break
            now = time.monotonic()
            if not all(os.path.exists(os.path.join(watch, f"boot_{r}")) for r in range(n)):
                if t_alive is None and all(os.path.exists(os.path.join(watch, f"alive_{r}")) for r in range(n)):
                    t_alive = now
                if t_alive is not None and now - t_alive > boot_timeout:
                    late = [r for r in range(n) if not os.path.exists(os.path.join(watch, f"boot_{r}"))]
                    reason = f"communicator bootstrap not finished {boot_timeout:.0f} s after all ranks were up (waiting for rank(s) {late})"
                    break
                if t_alive is None and now - t0 > start_timeout:
                    late = [r for r in range(n) if not os.path.exists(os.path.join(watch, f"alive_{r}"))]
                    reason = f"rank(s) {late} not up after {start_timeout:.0f} s"
                    break
            if now - t0 > run_timeout:
                reason = f"run not finished after {run_timeout:.0f} s"
                break
            time.sleep(0.1)
        if reason is not None:
            _kill_ranks(procs)
        line = None
        if reason is None:
            lines = [ln for ln in open(os.path.join(watch, "rank0.stdout"), "rb").read().decode(errors="replace").splitlines() if ln.startswith("{")]
            if lines:
                line = lines[-1]
            else:
                reason = "rank 0 finished without a JSON line"
        return line, reason, time.monotonic() - t0
    finally:
        _kill_ranks(procs)
        shutil.rmtree(watch, ignore_errors=True)


def launch_ranks(args, argv):
    """parent of a self-launched multi-rank run (never initialises the GPU); returns the exit status"""
    boot_timeout = float(os.environ.get("S3_BENCH_BOOT_TIMEOUT_S", "120"))
    run_timeout = float(os.environ.get("S3_BENCH_RUN_TIMEOUT_S", "1500"))
    start_timeout = float(os.environ.get("S3_BENCH_START_TIMEOUT_S", "400"))     # (the first `import torch` of a fresh box: 1-2 minutes)
    attempts = []
    plans = [dict()] if os.environ.get("S3_DIST_BACKEND") == "gloo" else [dict(), dict(S3_DIST_BACKEND="gloo")]
    for extra in plans:
        backend = extra.get("S3_DIST_BACKEND", os.environ.get("S3_DIST_BACKEND", "rccl"))
        line, reason, secs = _run_ranks(args.gpus, argv, extra, boot_timeout, run_timeout, start_timeout)
        attempts.append(dict(backend=backend, seconds=round(secs, 1), ok=reason is None, **({} if reason is None else {"failure": reason})))
        if reason is None:
            try:
                res = json.loads(line)
                res["launcher"] = dict(self_launched=True, ranks=args.gpus, attempts=attempts,
                                       note="bench.py --gpus N without WORLD_SIZE: a parent that never touches the GPU starts N fresh "
                                            "ranks and watches them")
                line = json.dumps(res)
            except ValueError:
                pass
            print(line, flush=True)
            return 0
        print(f"[bench] attempt on {backend} failed: {reason}; all ranks killed", file=sys.stderr, flush=True)
    print(json.dumps({"metric": "Mcells*snapshots/s interpolated", "value": None, "unit": "Mcells*snapshots/s", "n_gpus": args.gpus,
                      "steps": args.steps, "warmup": args.warmup, "error": "; ".join(f"{a['backend']}: {a.get('failure')}" for a in attempts),
                      "launcher": dict(self_launched=True, ranks=args.gpus, attempts=attempts)}), flush=True)
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cylinder3D_Re3900", choices=sorted(WORKLOADS) + ["svd"])
    ap.add_argument("--t-batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baselines, the device-resident and the end-to-end legs (profiling runs)")
    ap.add_argument("--no-batches", action="store_true", help="skip the roofline_batches sub-records")
    ap.add_argument("--no-pitched-copy", action="store_true", help="skip the pitched_copy sub-record (profiling runs: its launches may "
                    "carry the headline kernel's name -- 16-byte aligned rows of up to four chunks -- and would be averaged into its statistics)")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes "
                    "before the GPU is touched); the recorded value of profiles/ is reported instead, labelled")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)      # (the short run a PMC pass profiles)
    ap.add_argument("--n-comp", type=int, default=1, help="components per snapshot: the row holds n_comp * t_batch values")
    ap.add_argument("--shard", choices=["cells", "snapshots"], default="cells",
                    help="N>1: every rank interpolates its contiguous range of the generated cells for the same snapshots "
                         "(leaf-cell shards, default) or all cells for its own snapshot batches")
    ap.add_argument("--direct", action="store_true", help="time the direct gather kernel (s3_interp) instead of the "
                    "planned LDS-tiled kernel (s3_interp_planned)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))            # no launcher around us: start (and watch) the N ranks ourselves
    if args.traffic_child:                 # what a PMC pass profiles: the headline launch only, three times
        args.steps, args.warmup = CHILD_LAUNCHES, 1
        args.no_cpu_baseline = args.no_batches = args.no_pitched_copy = args.no_traffic = True
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        ap.error(f"--gpus {args.gpus} but WORLD_SIZE={world}: either start `python bench.py --gpus {args.gpus}` without a launcher (it "
                 f"starts its own ranks) or launch N ranks with `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
                 f"--master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")

    # stdout carries exactly one JSON line: libraries that print there (the RCCL version banner at communicator
    # creation) are sent to stderr for the whole run, the result goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    _mark("start")                         # (watchdog protocol of a self-launched run: nothing has touched the GPU yet)
    if rank == 0:
        query_rocm_smi()                   # (before anything initialises the GPU)
    measured = None
    if rank == 0 and world == 1 and not args.no_traffic and args.workload != "svd" and not args.direct:
        # (also before anything initialises the GPU: the passes are child processes of this one)
        child_argv = [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline", "--no-batches", "--no-pitched-copy")]
        measured = optional_leg("measure_traffic", lambda: measure_traffic(child_argv, "interp_planned"))
        if isinstance(measured, dict) and "error" in measured:
            measured = None
    # S3_BENCH_SHARE_GPU=1 + S3_DIST_BACKEND=gloo: rehearsal of the N>1 code path on a box with a single GPU
    share = os.environ.get("S3_BENCH_SHARE_GPU") == "1"
    pt.cuda.set_device(0 if share else local_rank)
    from sparsespatialsampling_amd import geometry, hipops, parallel
    from sparsespatialsampling_amd.s_cube import SamplingTree
    _mark("alive")                         # (watchdog protocol of a self-launched run; no-ops otherwise)
    if os.environ.get("S3_DIST_BACKEND") == "gloo" and world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    comm = parallel.init()                 # RCCL communicator inside libs3hip.so when world > 1
    _mark("boot")
    import logging
    logging.getLogger().setLevel(logging.WARNING)
    if args.workload == "svd":             # the downstream consumer: its own line, its own roofline
        if world != 1:
            ap.error("--workload svd is a single-GPU measurement")
        bench_svd(args, json_fd)
        parallel.shutdown()
        return

    cfg = dict(WORKLOADS[args.workload])
    if args.t_batch:
        cfg["t_batch"] = args.t_batch
    k = 26
    x, metric, geos, tree_kw = build_case(args.workload, cfg, geometry)

    # ---- refine (grid generation), timed separately -----------------------------------------------------------
    # one tiny run first: context creation, the load of the library's code objects and the first launch of every kernel
    # are one-off start-up cost of the process, not of a grid generation
    rng0 = np.random.default_rng(0)
    xs = rng0.random((20000, 3))
    warm = SamplingTree(pt.from_numpy(xs), pt.from_numpy(0.1 + np.exp(-4 * np.linalg.norm(xs - 0.5, axis=1))),
                        [geometry.CubeGeometry("domain", True, [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]),
                         geometry.CylinderGeometry3D("c", False, [(0.5, 0.5, -1.0), (0.5, 0.5, 2.0)], 0.1, refine=True)],
                        uniform_level=3, min_metric=0.3)
    warm.refine()
    warm.close()
    del warm, xs
    # the grid generation is timed four times: the first full-size run of the process also pays for the device allocations
    # (cell arrays, topology tables, allocator pools) -- it is reported apart, `refine_wall_s` is the median of the other three
    timings = []
    for attempt in range(1 if args.traffic_child else 4):
        pt.cuda.synchronize()
        comm.barrier()
        t0 = time.perf_counter()
        tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **tree_kw)
        t_init = time.perf_counter() - t0
        tree.refine()
        pt.cuda.synchronize()
        timings.append((comm.allreduce_max(time.perf_counter() - t0), t_init))
        centers = tree.all_centers.numpy()
        tree_out = (tree.all_centers, tree.all_nodes, tree.face_ids, tree.all_levels, float(tree.width))
        sha_gpu = grid_sha(tree.all_centers, tree.all_levels, tree.face_ids, tree.all_nodes) if attempt == 3 else None
        if args.traffic_child:
            timings = timings * 4
        info = dict(tree.data_final_mesh)
        n_cells_total = tree._topo_engine.n_created
        tree.close()
        del tree
    refine_first_s = timings[0][0]
    refine_s = float(np.median([t[0] for t in timings[1:]]))            # median of three steady runs
    t_init = float(np.median([t[1] for t in timings[1:]]))

    # ---- KNN cache (once) -------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    nc_total = len(centers)
    knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
    my_centers, shard_counts = centers, [nc_total]
    if world > 1 and args.shard == "cells":
        # this rank's share of the generated leaf cells: a spatially compact blob, cut so that every rank moves the same
        # number of bytes per snapshot (parallel.LeafShards, what ExportData does with several ranks)
        shards = parallel.LeafShards(knn, centers, k, rank, world, comm)
        my_centers, shard_counts = np.ascontiguousarray(centers[shards.mine]), shards.counts
        del shards
    idx, dist_ = knn.query(my_centers, k)
    w = hipops.idw_weights(dist_)
    knn.close()
    del dist_
    # only the source rows this rank's cells reference are resident (ExportData uploads exactly those per batch)
    # (in Hilbert order of their coordinates, as ExportData keeps them; S3_BENCH_ROW_ORDER=id: ascending point id)
    used, remap = hipops.referenced_rows([idx], len(x), coords=None if os.environ.get("S3_BENCH_ROW_ORDER") == "id" else x)
    n_rows = int(used.numel())
    hipops.remap_indices(idx, remap)
    del remap
    plan = None
    if not args.direct:
        plan = hipops.InterpPlan(idx, n_rows, my_centers, tile_cells=int(os.environ.get("S3_TILE_CELLS", "0")))
        plan.set_weights(w)
        plan.set_source_ids(used.contiguous(), len(x))       # what ExportData does for device-resident batches
    pt.cuda.synchronize()
    knn_cache_s = time.perf_counter() - t0
    nc, t_b = len(my_centers), cfg["t_batch"]
    row_len = t_b * args.n_comp                      # values per source row: [N, n_comp, T] flattened

    # ---- synthetic snapshot batch resident in HBM, as the API receives it ----------------------------------------
    # dense [N_points, n_comp * T] fp32: a row for EVERY point of the CFD mesh (interpolate_data / ExportData.export get the
    # field like this, export.py:128-167, 446-468); the kernels read the rows the grid references where they lie
    gen = pt.Generator(device="cuda").manual_seed(1234 + (rank if args.shard == "snapshots" else 0))
    data = pt.empty((len(x), row_len), dtype=pt.float32, device="cuda")
    data.normal_(generator=gen)
    out = pt.empty((nc, row_len), dtype=pt.float64, device="cuda")
    idx_full = used.long()[idx.long()].to(pt.int32) if args.direct else None       # --direct: ids in the full table

    def step():
        if plan is not None:
            plan.interp_src(data, out=out)
        else:
            hipops.interp(w, idx_full, data, out=out)

    _mark("run")
    for _ in range(args.warmup):
        step()
    pt.cuda.synchronize()
    comm.barrier()
    ev = [(pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    pt.cuda.synchronize()
    comm.barrier()
    elapsed = comm.allreduce_max(time.perf_counter() - t0)
    launch_ms = [a.elapsed_time(b) for a, b in ev]                       # HIP events on the launch stream, per launch
    kernel_ms = float(np.mean(launch_ms))

    # the same launch on the pitched, compacted copy of the referenced rows (Hilbert order, whole 128-byte lines per row:
    # the layout ExportData uploads HOST batches into -- rounds 1-3 quoted the headline on it)
    pitched = None
    if rank == 0 and world == 1 and plan is not None and not args.no_pitched_copy:
        def pitched_leg():
            rows_p = hipops.gather_rows(data, used.contiguous(), hipops.padded_rows(n_rows, row_len, pt.float32, "cuda",
                                                                                   int(os.environ.get("S3_BENCH_PITCH_EXTRA", "0"))))
            pms = launch_times_ms(lambda: plan.interp(w, rows_p, out=out), args.steps, args.warmup)
            return dict(layout=f"the {n_rows} referenced rows only, Hilbert order, pitch {int(rows_p.stride(0)) * 4} B (whole 128-byte lines)",
                        kernel=planned_kernel_name(row_len, k, plan.n_tiles), **ms_stats(pms))
        pitched = optional_leg("pitched_copy", pitched_leg)
        step()                                       # `out` holds the headline's result again

    # where the caller's batch LIES in HBM moves the same launch by 3-4 % (round 5, tools/placement_probe.py: four 20-GB allocations of
    # one process, 0.466 / 0.481 / 0.482 / 0.481 of peak in the same minute -- the boxes' "faster and slower states" are this): the
    # headline stays the allocation a caller gets, the same launch on a SECOND allocation of the same batch is recorded beside it
    placement = None
    if rank == 0 and world == 1 and plan is not None and not args.no_pitched_copy and not args.traffic_child:
        def placement_leg():
            ref = out.clone()                         # the headline's result
            other = pt.empty_like(data)
            other.copy_(data)
            ms2 = launch_times_ms(lambda: plan.interp_src(other, out=out), args.steps, args.warmup)
            same = bool(pt.equal(out, ref))
            del other, ref
            return dict(note="the same launch with the batch in a second allocation of the same size (both resident at the time)",
                        same_bits_as_the_headline=same, **ms_stats(ms2))
        placement = optional_leg("placement", placement_leg)
        step()

    # N > 1: the product's export path with N ranks (every rank takes part; after the timed region of the headline)
    sharded_leg = None
    if world > 1 and args.shard == "cells" and plan is not None and os.environ.get("S3_BENCH_NO_EXPORT_LEG") != "1":
        del data
        pt.cuda.empty_cache()
        # (the box workload in ITS batches: 16 snapshots per field and batch)
        t_leg = 16 if cfg.get("kind") == "box" else 200
        sharded_leg = optional_leg("export_sharded", lambda: export_sharded(x, centers, k, comm, t=t_leg))    # (a failure on ALL ranks: no line lost)
        data = None

    if args.traffic_child and plan is not None and world == 1:
        # (PMC child pass: the batch shapes behind the headline, in the order measure_traffic() attributes the launches by)
        del data
        pt.cuda.empty_cache()
        for _, rl, _ in BATCH_SHAPES:
            table = pt.empty((len(x), rl), dtype=pt.float32, device="cuda").normal_(generator=gen)
            o = pt.empty((nc, rl), dtype=pt.float64, device="cuda")
            for _ in range(CHILD_LAUNCHES):
                plan.interp_src(table, out=o)
            rows_p = hipops.gather_rows(table, used.contiguous(), hipops.padded_rows(n_rows, rl, pt.float32, "cuda"))
            for _ in range(CHILD_LAUNCHES):
                plan.interp(w, rows_p, out=o)
            pt.cuda.synchronize()
            del table, o, rows_p
        data = None
    copy_bw = copy_bandwidth_gbs() if rank == 0 else None
    yard = None
    if rank == 0 and world == 1 and plan is not None and data is not None and not args.traffic_child and not args.no_pitched_copy:
        yard = optional_leg("yardsticks", lambda: yardsticks(hipops, plan, data, args.steps, args.warmup))
    if rank == 0:
        units = (nc * world if args.shard == "snapshots" else nc_total) * t_b * args.steps
        value = units / elapsed / 1e6
        # algorithmic HBM bytes of one launch on this rank: every referenced source row once + every output
        # once + idx (int32) / weights (f64) once
        b_alg = n_rows * row_len * 4 + nc * row_len * 8 + nc * k * (4 + 8)
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        workload = (f"{args.workload} (synthetic, SURVEY 8(d)): dense batch [{len(x)} points, {row_len}] fp32 resident in HBM as "
                    f"interpolate_data receives it, read in place ({n_rows} rows referenced) x {t_b} snapshots per step, "
                    f"{nc_total} generated cells, k={k}, fp32 in / f64 out")
        shape_key = f"T{t_b}" + (f"x{args.n_comp}" if args.n_comp > 1 else "")
        traffic, traffic_src, traffic_stale = (recorded_traffic(f"{args.workload}/{shape_key}/inplace") if plan is not None and world == 1
                                               else (None, None, None))
        recorded = traffic
        traffic_how = None if traffic is None else f"recorded, not measured in this run: {traffic_src}"
        if measured is not None:           # measured in THIS run (two PMC child passes before the GPU was touched): preferred
            traffic, traffic_how, traffic_stale = measured["traffic"], measured["how"], False
        res = {
            "metric": "Mcells*snapshots/s interpolated", "value": value, "unit": "Mcells*snapshots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.shard == "snapshots" else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "n_points": len(x), "resident_source_rows": n_rows, "n_cells": nc_total,
                       "t_batch": t_b, "k": k, "n_comp": args.n_comp, "shape_key": shape_key, "input": "dense device tensor, read in place",
                       "parallelism": f"{'snapshot-axis' if args.shard == 'snapshots' else 'leaf-cell'} shards x{world}",
                       "cells_per_rank": shard_counts, "collectives": comm.name},
            "device": device_identity(), "code_sha": code_sha(),
            "refine_wall_s": refine_s, "refine_init_s": t_init, "refine_first_run_wall_s": refine_first_s,
            "refine_runs_s": [t[0] for t in timings],
            "refine_iterations": info["iterations"],
            "refine_cells_created": n_cells_total, "refine_leaves_per_s": nc_total / refine_s, "knn_cache_s": knn_cache_s,
            "captured_metric": info["metric_per_iter"][-1], "grid_sha256": sha_gpu,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "copy_kernel_GBs": copy_bw, "frac_of_copy_kernel": achieved / copy_bw,
                         "traffic": traffic, "traffic_source": traffic_how, "traffic_stale": traffic_stale,
                         "traffic_over_algorithmic": None if traffic is None else traffic / b_alg,
                         "traffic_measured": measured, "traffic_recorded_in_profiles": recorded,
                         "kernel": "interp_kernel<float,4>" if plan is None else planned_kernel_name(row_len, k, plan.n_tiles, row_len, table_bytes=len(x) * row_len * 4),
                         "staged_rows_per_launch": None if plan is None else plan.total_rows, **ms_stats(launch_ms),
                         "frac_best_launch": b_alg / (min(launch_ms) * 1e-3) / 8e12,
                         "algorithmic_bytes": b_alg, "resident_source_rows": n_rows, "cells_on_this_rank": nc,
                         "gather_upper_bound_bytes": nc * k * row_len * 4 + nc * row_len * 8},
        }
        if yard is not None:
            res["roofline"]["yardsticks"] = yard
            mix = yard.get("mix_7r2w", {}).get("GBs", 0.0)
            if mix > 0.0 and traffic is not None:
                # the launch's measured bytes per second over what the hand-written 78 % / 22 % streaming mix moves per second
                res["roofline"]["streaming_ceiling_GBs"] = mix
                res["roofline"]["traffic_GBs"] = traffic / (kernel_ms * 1e-3) / 1e9
                res["roofline"]["frac_of_streaming_ceiling"] = traffic / (kernel_ms * 1e-3) / 1e9 / mix
                res["roofline"]["frac_if_streaming_ceiling_were_reached"] = b_alg / traffic * mix / 8000.0
            loads = yard.get("plan_loads_one_line_per_visit", {})
            if "kernel_ms" in loads:
                res["roofline"]["plan_loads_over_launch"] = loads["kernel_ms"] / kernel_ms
        if placement is not None:
            if "kernel_ms" in placement:
                placement["frac"] = b_alg / (placement["kernel_ms"] * 1e-3) / 8e12
                placement["second_over_first_allocation"] = placement["kernel_ms"] / kernel_ms
            res["roofline"]["second_allocation"] = placement
        if pitched is not None:
            if "kernel_ms" in pitched:
                pitched["frac"] = b_alg / (pitched["kernel_ms"] * 1e-3) / 8e12
                pitched["in_place_over_pitched"] = kernel_ms / pitched["kernel_ms"]
            res["roofline"]["pitched_copy"] = pitched
        if sharded_leg is not None:
            res["export_sharded"] = sharded_leg
        if world == 1 and plan is not None and not args.no_batches:
            # the batch lengths the reference exports with (examples/s3_for_cylinder3D_Re3900.py:28-69, utils.py:204-226)
            key = args.workload
            res["roofline_batches"] = {name: optional_leg(f"roofline_batches.{name}", lambda rl=rl, label=label, name=name: batch_record(
                                                 hipops, plan, w, used.contiguous(), len(x), nc, k, rl, label, f"{key}/{name}",
                                                 args.steps, args.warmup, gen, measured))
                                       for name, rl, label in BATCH_SHAPES if rl != row_len}
            if "kernel_ms" in res["roofline_batches"].get("T25", {}):
                res["roofline_batches"]["T25"]["numbering_follows_space"] = optional_leg("numbering_follows_space", lambda: numbering_follows_space(
                    hipops, x, idx, used, w, my_centers, k, 25, args.steps, args.warmup, gen))
        if not args.no_cpu_baseline and world == 1:       # reported at N=1 only
            bare = {t_b: kernel_ms}
            if "kernel_ms" in res.get("roofline_batches", {}).get("T25", {}):
                bare[25] = res["roofline_batches"]["T25"]["kernel_ms"]
            # the transport legs BEFORE the CPU baselines: the download buffers of a 1000-snapshot batch (2 x 3.7 GB of page-locked
            # memory) allocated after the CPU baseline had run came down at 31 GB/s instead of 57 (cause not isolated) -- on
            # every box tried, whatever the threads' placement (device_resident_input.T1000.fit_data_ms 118 ms against 66 in a process
            # that does only that; HISTORY 6.2)
            if not cfg.get("kind") == "box":
                res["device_resident_input"] = optional_leg("device_resident_input", lambda: device_resident_input(x, centers, k, sorted({25, t_b}), bare))
                pt.cuda.empty_cache()
                res["end_to_end"] = optional_leg("end_to_end", lambda: end_to_end(x, centers, k))
                res["end_to_end"]["export_to_file"] = optional_leg("export_to_file", lambda: export_to_file(x, metric, tree_out, k))
            # (required by the contract, but it must not cost the line either: a host short of memory for the 9.7 GB of referenced
            # rows falls back to a bounded slice of the cells, and only a failure of that too leaves an error record)
            res["cpu_baseline"] = optional_leg("cpu_baseline", lambda: cpu_baseline(w, idx, data, k, used))
            if "value" not in res["cpu_baseline"]:
                first_error = res["cpu_baseline"]["error"]
                os.environ["S3_BENCH_CPU_CELLS"] = "100000"
                res["cpu_baseline"] = optional_leg("cpu_baseline", lambda: cpu_baseline(w, idx, data, k, used))
                res["cpu_baseline"]["all_cells_failed_with"] = first_error
            del data, out
            pt.cuda.empty_cache()

            def refine_leg():
                rcb = refine_cpu_baseline(args.workload, x, metric, geos, tree_kw, nc_total, sha_gpu)
                rcb.update(gpu_wall_s=refine_s, gpu_runs_s=[t[0] for t in timings[1:]], speedup=rcb["cpu_wall_s"] / refine_s)
                return rcb
            res["refine_cpu_baseline"] = optional_leg("refine_cpu_baseline", refine_leg)
            if "value" in res["cpu_baseline"]:
                cpu_g = res["cpu_baseline"]["value"] / 1e3     # G cell*snapshots/s of the CPU port
                ratios = {"in_hbm": value / 1e3 / cpu_g, "cpu_port_Gcells_snapshots_per_s": cpu_g,
                          "note": "GPU rate / rate of the OpenMP oracle port on this box's host cores; in_hbm: the headline (dense batch "
                                  "resident, read in place), device_resident: the same through ExportData._upload + neighbour table, "
                                  "host_to_host: end_to_end"}
                if f"T{t_b}" in res.get("device_resident_input", {}):
                    ratios["device_resident"] = res["device_resident_input"][f"T{t_b}"]["Gcells_snapshots_per_s"] / cpu_g
                e2e = res.get("end_to_end", {})
                timed = [n for n in e2e if n.startswith("T") and "Gcells_snapshots_per_s" in e2e[n]]
                for name in timed:
                    e2e[name]["speedup_vs_cpu_port"] = e2e[name]["Gcells_snapshots_per_s"] / cpu_g
                if timed:
                    ratios["host_to_host"] = max(e2e[n]["speedup_vs_cpu_port"] for n in timed)
            else:
                ratios = {"error": "no CPU baseline in this run"}
            res["gpu_over_cpu"] = ratios
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    comm.barrier()
    parallel.shutdown()
    if os.environ.get("S3_DIST_BACKEND") == "gloo" and world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
