"""worker of test_sharded_export_matches_single_rank (tests/test_gpu_fuzz.py): grid generation + ExportData.export() of a
scalar field in two batches and a vector field, cell centres and vertices, run by 1 or W processes (SPMD; W > 1: leaf-cell
shards, rank 0 writes).  argv: output directory."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch as pt


def main(out_dir):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")            # S3_DIST_BACKEND=gloo: the exchange steps of ranks that share one GPU
    pt.cuda.set_device(0)
    from sparsespatialsampling_amd import geometry, parallel
    from sparsespatialsampling_amd.export import ExportData
    from sparsespatialsampling_amd.sparse_spatial_sampling import SparseSpatialSampling
    from inputs import refine_inputs
    comm = parallel.init()
    assert (comm.rank, comm.world) == (rank, world)
    x, y, geos, kw = refine_inputs("refine_3d_metric", geometry)
    s3 = SparseSpatialSampling(pt.from_numpy(x), pt.from_numpy(y), geos, out_dir, "case", uniform_levels=kw["uniform_level"],
                               min_metric=kw["min_metric"])
    s3.execute_grid_generation()                              # every rank generates the grid, rank 0 writes its files
    n_t = 11
    times = [f"{0.1 * i:.1f}" for i in range(n_t)]
    rng = np.random.default_rng(21)
    p = rng.standard_normal((len(x), 1, n_t)).astype(np.float32)
    u = rng.standard_normal((len(x), 3, n_t))                 # float64 rows
    ex = ExportData(s3, write_times=times, interpolate_at_vertices=True)
    for a, b in ((0, 6), (6, 11)):
        ex.export(pt.from_numpy(x), pt.from_numpy(p[:, :, a:b]), "p", n_snapshots_total=n_t)
    ex.export(pt.from_numpy(x), pt.from_numpy(u), "U")
    if world > 1:
        table = ex._table_centers
        assert table.shard is not None and sum(table.shard.counts) == len(s3.centers)
        n_rows = len(x) if ex._used_rows is None else int(ex._used_rows.numel())
        print(f"rank {rank}: {len(table.shard.mine)} of {len(s3.centers)} cells, {n_rows} of {len(x)} source rows", flush=True)
    comm.barrier()
    parallel.shutdown()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    print("worker ok", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
