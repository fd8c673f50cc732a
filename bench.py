#!/usr/bin/env python3
"""
bench.py -- S^3 hot path on MI355X: snapshot interpolation throughput (+ refine wall-clock) on the synthetic
cylinder3D_Re3900 workload of BASELINE.json / SURVEY.md 8(d).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A *step* is one pass of the interpolation hot path (`export.interpolate_data` -> s3_interp) over one batch of synthetic
snapshots: data [N_points, 1, T_batch] fp32 already resident in HBM -> out [N_cells, 1, T_batch] f64 in HBM, with the
KNN indices/weights of the generated grid resident as well.  The grid itself comes from `SamplingTree.refine()` run on
the GPU before the timed region; its wall-clock is reported as `refine_wall_s` in the same JSON line.

Multi-GPU (one process per GPU, launched by torch.distributed.run): the path shards over the snapshot axis -- every
rank holds the (replicated) grid + KNN cache and interpolates its own snapshot batches, no data-path collective
("scaling": "weak").  refine() runs replicated on every rank with the captured-metric reduction split across ranks
(one 8-byte RCCL all-reduce per refine iteration).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch as pt
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_points, T_batch, uniform_levels, min_metric)
    "cylinder3D_Re3900": dict(n=5_000_000, lo=[0.0, 0.0, 0.0], hi=[2.4, 2.0, 0.1 * np.pi], t_batch=1000,
                              uniform_levels=5, min_metric=0.75, seed=2),
    "cylinder3D_small": dict(n=300_000, lo=[0.0, 0.0, 0.0], hi=[2.4, 2.0, 0.1 * np.pi], t_batch=256,
                             uniform_levels=4, min_metric=0.6, seed=2),
    # HBM stress -- random centroids in the unit box, cell budget, batches of 16 snapshots (64-byte rows)
    "box5e7": dict(kind="box", n=50_000_000, t_batch=16, uniform_levels=5, n_cells_max=10_000_000, seed=3),
    "box5e7_small": dict(kind="box", n=5_000_000, t_batch=16, uniform_levels=4, n_cells_max=1_000_000, seed=3),
}


def synthetic_box(cfg):
    """uniform random centroids in the unit box, smooth metric peaking at the centre with short waves on top (C4)"""
    rng = np.random.default_rng(cfg["seed"])
    x = rng.random((cfg["n"], 3))
    r = np.sqrt(((x - 0.5) ** 2).sum(1))
    metric = 0.05 + np.exp(-6 * r) * (1 + 0.5 * np.sin(25 * x[:, 0]) * np.cos(17 * x[:, 1]))
    return x, metric


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


def cpu_baseline(x_host, centers, k, t_sample, seconds=12.0):
    """the CPU oracle (C + OpenMP restatement of export.py:446-468) on a bounded sample of the same workload"""
    from oracle import s3_oracle as orc
    nc = min(len(centers), 10_000)
    n_src = min(len(x_host), 200_000)          # brute-force KNN in the oracle: keep the index build of the sample cheap
    idx, dist = orc.knn(x_host[:n_src], centers[:nc], k)
    w = orc.idw_weights(dist)
    data = np.random.default_rng(0).standard_normal((n_src, 1, t_sample)).astype(np.float32)
    orc.interp(w, idx, data)                   # warm
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.interp(w, idx, data)
        reps += 1
    dt = time.perf_counter() - t0
    return dict(value=nc * t_sample * reps / dt / 1e6, unit="Mcells*snapshots/s", cores=orc.num_threads(), kind="port",
                sample=f"oracle/s3_oracle.c s3o_interp (OpenMP), {nc} cells x {t_sample} snapshots x {reps} passes, "
                       f"k={k}, fp32 in / f64 out, {n_src} source points")


def recorded_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/rNN/summary.json:
    FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, separate rocprofv3 --pmc runs of this very
    command); None when no pass was recorded for this workload string"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "summary.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except (OSError, ValueError):
            continue
        if rec.get("workload") == workload and "traffic_bytes_per_launch" in rec:
            return rec["traffic_bytes_per_launch"]
    return None


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cylinder3D_Re3900", choices=sorted(WORKLOADS))
    ap.add_argument("--t-batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", choices=["snapshots", "cells"], default="snapshots",
                    help="N>1: every rank interpolates all cells for its own snapshot batches (weak scaling, default) or "
                         "its contiguous range of the generated cells for the same snapshots (strong scaling)")
    ap.add_argument("--direct", action="store_true", help="time the direct gather kernel (s3_interp) instead of the "
                    "planned LDS-tiled kernel (s3_interp_planned)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print there (the RCCL version banner at communicator
    # creation) are sent to stderr for the whole run, the result goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # S3_BENCH_SHARE_GPU=1 + S3_DIST_BACKEND=gloo: rehearsal of the N>1 code path on a box with a single GPU
    share = os.environ.get("S3_BENCH_SHARE_GPU") == "1"
    pt.cuda.set_device(0 if share else local_rank)
    # S3_BENCH_FORCE_DIST=1: run the process-group code path (RCCL init, barriers, MAX all-reduce) with a single rank too
    use_dist = world > 1 or os.environ.get("S3_BENCH_FORCE_DIST") == "1"
    if use_dist:
        backend = os.environ.get("S3_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=pt.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from sparsespatialsampling_amd import geometry, hipops
    from sparsespatialsampling_amd.s_cube import SamplingTree
    import logging
    logging.getLogger().setLevel(logging.WARNING)

    cfg = dict(WORKLOADS[args.workload])
    if args.t_batch:
        cfg["t_batch"] = args.t_batch
    k = 26
    x, metric = synthetic_cylinder3d(cfg)
    if os.environ.get("S3_BENCH_MESH_ORDER") == "1":
        # experiment: points stored in a spatially coherent order (as a CFD mesh numbering would be) instead of random
        q = ((x - x.min(0)) / (x.max(0) - x.min(0)).max() * 1023).astype(np.int64)

        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
        x, metric = np.ascontiguousarray(x[order]), np.ascontiguousarray(metric[order])
    geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
            geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]

    # ---- refine (grid generation), timed separately -----------------------------------------------------------
    # one tiny call first: context creation and the load of the library's code objects are one-off start-up cost
    warm = hipops.KnnIndex(np.random.default_rng(0).random((64, 3)))
    warm.query(np.zeros((1, 3)), 4)
    warm.close()
    pt.cuda.synchronize()
    t0 = time.perf_counter()
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"],
                        min_metric=cfg["min_metric"])
    tree.refine()
    pt.cuda.synchronize()
    refine_s = time.perf_counter() - t0
    centers = tree.all_centers.numpy()
    info = dict(tree.data_final_mesh)
    n_cells_total = tree._topo.n_cells
    tree._backend.close()
    del tree

    # ---- KNN cache (once) -------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3))
    idx, dist_ = knn.query(centers, k)
    w = hipops.idw_weights(dist_)
    pt.cuda.synchronize()
    knn_cache_s = time.perf_counter() - t0
    nc_total = len(centers)
    if world > 1 and args.shard == "cells":
        # leaf cells shard across ranks: contiguous ranges of the generated grid, no data-path collective
        from sparsespatialsampling_amd.parallel import shard_range
        c0, c1 = shard_range(nc_total, rank, world)
        centers, idx, w = centers[c0:c1], idx[c0:c1].contiguous(), w[c0:c1].contiguous()
    knn.close()
    del dist_
    plan = None
    if not args.direct and cfg["t_batch"] % 4 == 0:      # the tiled kernel needs 16-byte aligned fp32 rows
        plan = hipops.InterpPlan(idx, len(x), centers, tile_cells=int(os.environ.get("S3_TILE_CELLS", "0")))
        pt.cuda.synchronize()
    knn_cache_s = time.perf_counter() - t0
    nc, n_src, t_b = len(centers), len(x), cfg["t_batch"]
    n_unique = int(pt.unique(idx).numel())          # plumbing: only used for the algorithmic byte count

    # ---- synthetic snapshot batch resident in HBM ---------------------------------------------------------------
    gen = pt.Generator(device="cuda").manual_seed(1234 + (rank if args.shard == "snapshots" else 0))
    if plan is not None:
        # same layout the export path uploads into: [N, n_comp*T] with the row pitch padded to a multiple of 128 bytes
        data = hipops.padded_rows(n_src, t_b, pt.float32, "cuda", int(os.environ.get("S3_BENCH_PITCH_EXTRA", "0")))
        data.normal_(generator=gen)
    else:
        data = pt.randn((n_src, 1, t_b), dtype=pt.float32, device="cuda", generator=gen)
    out = pt.empty((nc, t_b) if plan is not None else (nc, 1, t_b), dtype=pt.float64, device="cuda")

    def step():
        if plan is not None:
            plan.interp(w, data, out=out)
        else:
            hipops.interp(w, idx, data, out=out)

    for _ in range(args.warmup):
        step()
    pt.cuda.synchronize()
    if use_dist:
        dist.barrier()
    ev = [(pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    pt.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = pt.tensor([elapsed], dtype=pt.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))       # HIP events on the launch stream

    copy_bw = copy_bandwidth_gbs() if rank == 0 else None
    if rank == 0:
        units = (nc * world if args.shard == "snapshots" else nc_total) * t_b * args.steps
        value = units / elapsed / 1e6
        # algorithmic HBM bytes of one launch: every referenced source row once + every output once +
        # idx (int32) / weights (f64) once
        b_alg = n_unique * t_b * 4 + nc * t_b * 8 + nc * k * (4 + 8)
        achieved = b_alg / (kernel_ms * 1e-3) / 1e9
        workload = (f"{args.workload} (synthetic, SURVEY 8(d) C3): {n_src} points x {t_b} snapshots "
                    f"per step, {nc_total} generated cells, k={k}, fp32 in / f64 out")
        res = {
            "metric": "Mcells*snapshots/s interpolated", "value": value, "unit": "Mcells*snapshots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if args.shard == "snapshots" else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "n_points": n_src, "n_cells": nc_total, "t_batch": t_b, "k": k, "n_comp": 1,
                       "parallelism": f"{'snapshot-axis' if args.shard == 'snapshots' else 'leaf-cell'} shards x{world}"},
            "refine_wall_s": refine_s, "refine_iterations": info["iterations"], "refine_cells_created": n_cells_total,
            "refine_leaves_per_s": nc / refine_s, "knn_cache_s": knn_cache_s,
            "captured_metric": info["metric_per_iter"][-1],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "copy_kernel_GBs": copy_bw, "frac_of_copy_kernel": achieved / copy_bw,
                         "traffic": recorded_traffic(workload) if plan is not None else None, "kernel": "interp_kernel<float,4>" if plan is None else "interp_planned_kernel<float>",
                         "staged_rows_per_launch": None if plan is None else plan.total_rows, "kernel_ms": kernel_ms,
                         "algorithmic_bytes": b_alg, "unique_source_rows": n_unique,
                         "gather_upper_bound_bytes": nc * k * t_b * 4 + nc * t_b * 8},
        }
        if not args.no_cpu_baseline and world == 1:       # reported at N=1 only
            res["cpu_baseline"] = cpu_baseline(x, centers, k, min(t_b, 64))
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
