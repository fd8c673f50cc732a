"""
Multi-GPU glue (one process per GPU, ``torch.distributed`` with the ``nccl`` backend = RCCL over xGMI).

The S^3 hot path shards without a data-path collective:

* **interpolation** -- the generated cells (or the snapshot axis) are split into contiguous per-rank ranges, each rank
  interpolates its range with the KNN index/weights of that range; outputs land in disjoint row ranges
  (``shard_range``).  No communication.
* **refine** -- every rank holds the replicated point cloud + cell arrays; the only reduction that spans all cells per
  iteration is the captured-metric numerator (sum of metric^2 over the leaves).  Each rank reduces a 1/W slice of the
  cell id range on its GPU and one 8-byte all-reduce combines them (``allreduce_sumsq``) -- the "one RCCL all-reduce
  per refinement iteration" of the north star.

With a single process both helpers degenerate to the local computation.
"""
import torch as pt
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n, rank=None, world_size=None):
    """contiguous, balanced [begin, end) slice of ``range(n)`` owned by ``rank``"""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(int(n), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_sumsq(backend, n_cells):
    """sum over leaf cells of metric^2; ranks reduce disjoint id ranges and all-reduce the partial sums"""
    rank, size = world()
    if size == 1:
        return backend.sumsq(n_cells)
    begin, end = shard_range(n_cells, rank, size)
    part = backend.sumsq_range(begin, end)          # 1-element device tensor (RCCL) or CPU tensor (gloo tests)
    if part.is_cuda and dist.get_backend() == "gloo":
        part = part.cpu()                           # single-GPU rehearsal of the multi-rank path
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    return float(part.item())
