"""
Multi-GPU layer of the S^3 path: one process per GPU, the collectives run inside libs3hip.so on RCCL over xGMI
(``s3_comm_*``, csrc/comm.hip).  SURVEY.md 8(e):

* **interpolation** -- the generated cells are split into contiguous per-rank ranges (``shard_range``); every rank builds
  the KNN cache / plan of its range, uploads only the source rows that range references and writes its own output rows.
  No collective.
* **refine** -- point cloud, KNN index and cell arrays are replicated; per batch every rank evaluates the KNN metric /
  gain of its 1/W slice of the new cells (``batch_slice``; the reference spreads the same work over a process pool,
  s_cube.py:207-241) and ONE grouped all-gather returns the slices to everybody.  The captured metric
  (s_cube.py:317-336) is reduced as partial sums over fixed 1024-cell blocks, every rank a share of the blocks, gathered
  and then added in block order on every rank: bit-identical for any number of ranks, so stopping decisions and therefore
  the grid cannot depend on the world size.

``get_comm()`` returns the process-wide communicator: ``RcclComm`` on GPUs (bootstrap: rank 0 creates the RCCL id and
publishes it through a ``torch.distributed.TCPStore`` on MASTER_ADDR:MASTER_PORT -- plumbing, no process group is
created), ``GlooComm`` for CPU tests / single-GPU rehearsals (``S3_DIST_BACKEND=gloo``, uses an initialised
``torch.distributed`` gloo group), ``SoloComm`` when there is one process.
"""
import ctypes as C
import os

import numpy as np
import torch as pt

SUMSQ_BLOCK = 1024           # cells per partial sum of the captured metric (include/s3hip.h S3_SUMSQ_BLOCK)


def shard_range(n, rank=None, world_size=None):
    """contiguous, balanced [begin, end) slice of ``range(n)`` owned by ``rank``"""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(int(n), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class LeafShards:
    """Cost-balanced, spatially compact shards of a set of target points (generated cell centres / vertices) for the
    interpolation.  Every rank computes the same partition: neighbour table of ALL targets (the KNN query
    is cheap next to one snapshot batch), its tile plan, tiles -- Hilbert order, so a run of tiles is a compact blob that
    shares few source rows with the other ranks -- cut into ``world`` runs of equal cost (bytes moved per snapshot:
    staged rows incl. halo + output rows).

    ``mine``      int64 host array, the targets of this rank (ascending ids)
    ``counts``    targets per rank; ``chunk`` = max(counts): slot size of the equal-chunk in-place all-gather
    ``slot_of``   device int32 [n]: target id -> row of the gathered ``[world * chunk, L]`` array
    """

    def __init__(self, knn, targets, k, rank, world_size):
        from . import hipops
        targets = hipops.to_device(targets, pt.float64)
        idx, _ = knn.query(targets, k)
        plan = hipops.InterpPlan(idx, knn.n, targets)
        order, cuts = plan.partition(world_size)
        plan.close()
        del idx
        self.rank, self.world, self.n = int(rank), int(world_size), int(targets.shape[0])
        if self.n < self.world:
            raise ValueError(f"{self.n} target points cannot be sharded over {self.world} ranks")
        if min(cuts[r + 1] - cuts[r] for r in range(self.world)) == 0:
            # fewer tiles than ranks (tiny grids): equal counts along the curve instead
            cuts = [shard_range(self.n, r, self.world)[0] for r in range(self.world)] + [self.n]
        self.counts = [cuts[r + 1] - cuts[r] for r in range(self.world)]
        self.chunk = max(self.counts)
        order_h = order.cpu().numpy()
        self.mine = np.sort(order_h[cuts[self.rank]:cuts[self.rank + 1]]).astype(np.int64)
        slot = np.empty(self.n, dtype=np.int32)
        for r in range(self.world):
            ids = np.sort(order_h[cuts[r]:cuts[r + 1]])
            slot[ids] = r * self.chunk + np.arange(len(ids), dtype=np.int32)
        self.slot_of = pt.from_numpy(slot).to(order.device)


def batch_slice(n, rank, world_size):
    """equal-sized chunks for an in-place all-gather: (chunk, begin, end) -- rank r owns [r * chunk, (r + 1) * chunk)
    clipped to n; the gathered array spans world_size * chunk entries"""
    chunk = -(-int(n) // int(world_size))
    begin = min(rank * chunk, n)
    return chunk, begin, min(begin + chunk, n)


class SoloComm:
    rank, world, name = 0, 1, "solo"

    def allgather_inplace(self, arrays, counts):
        pass

    def allreduce_max(self, value):
        return float(value)

    def barrier(self):
        pass

    def close(self):
        pass


class GlooComm(SoloComm):
    """``torch.distributed`` gloo group (CPU): the N > 1 protocol without GPUs, or several ranks sharing one GPU"""
    name = "gloo"

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather_inplace(self, arrays, counts):
        for a, cnt in zip(arrays, counts):
            if cnt == 0:
                continue
            t = pt.from_numpy(a) if isinstance(a, np.ndarray) else a
            flat = t.reshape(-1)[:self.world * cnt]
            host = flat.cpu() if flat.is_cuda else flat
            mine = host[self.rank * cnt:(self.rank + 1) * cnt].clone()
            out = pt.empty_like(host)
            self._dist.all_gather_into_tensor(out, mine)
            flat.copy_(out)

    def allreduce_max(self, value):
        t = pt.tensor([float(value)], dtype=pt.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self._dist.barrier()


class RcclComm(SoloComm):
    """communicator inside libs3hip.so (RCCL); device tensors only"""
    name = "rccl"

    def __init__(self, rank, world_size, store):
        from . import _lib, hipops
        self._lib, self._ops = _lib, hipops
        self.rank, self.world = int(rank), int(world_size)
        lib = _lib.hip_lib()
        ident = (C.c_char * 128)()
        if self.rank == 0:
            hipops.check(lib.s3_comm_unique_id(ident, 128), "s3_comm_unique_id")
            store.set("s3_comm_id", bytes(ident.raw))
        else:
            ident.raw = store.get("s3_comm_id")
        self._h = C.c_void_p(0)
        hipops.device()
        hipops.check(lib.s3_comm_init(ident, 128, self.rank, self.world, C.byref(self._h)), "s3_comm_init")
        self._scalar = pt.zeros(1, dtype=pt.float64, device=hipops.device())

    def allgather_inplace(self, arrays, counts):
        """``arrays[i]`` is a contiguous device tensor whose first ``world * counts[i]`` elements are gathered in place:
        rank r contributes elements [r * counts[i], (r + 1) * counts[i]).  One RCCL group for all arrays."""
        n = len(arrays)
        ptrs = (C.c_void_p * n)(*[a.data_ptr() for a in arrays])
        sizes = (C.c_size_t * n)(*[int(c) * a.element_size() for a, c in zip(arrays, counts)])
        for a, c in zip(arrays, counts):
            if not (a.is_cuda and a.is_contiguous() and a.numel() >= self.world * c):
                raise ValueError("allgather_inplace: contiguous device tensors with room for world * count elements required")
        self._ops.check(self._lib.hip_lib().s3_comm_allgather_inplace(self._h, ptrs, sizes, n, self._ops._stream()),
                        "s3_comm_allgather_inplace")

    def _allreduce(self, value, op):
        self._scalar[0] = float(value)
        self._ops.check(self._lib.hip_lib().s3_comm_allreduce_f64(self._h, C.c_void_p(self._scalar.data_ptr()), 1, op,
                                                                  self._ops._stream()), "s3_comm_allreduce_f64")
        return float(self._scalar.item())

    def allreduce_max(self, value):
        return self._allreduce(value, 1)

    def barrier(self):
        self._allreduce(0.0, 0)

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.hip_lib().s3_comm_destroy(h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass


_comm = None


def init(backend=None):
    """create the process-wide communicator from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT);
    idempotent.  ``backend``: "rccl" (default with more than one rank), "gloo" (needs an initialised torch.distributed
    group), None = S3_DIST_BACKEND or "rccl"."""
    global _comm
    if _comm is not None:
        return _comm
    rank, world_size = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = backend or os.environ.get("S3_DIST_BACKEND", "rccl")
    if backend == "gloo":
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            _comm = GlooComm() if dist.get_world_size() > 1 else SoloComm()
        else:
            _comm = SoloComm()
    elif world_size > 1 or os.environ.get("S3_COMM_FORCE") == "1":
        # the launcher's rendezvous store (torch.distributed.run hosts it itself and tells the workers so through
        # TORCHELASTIC_USE_AGENT_STORE; a bare environment gets a TCPStore served by rank 0): torch's own env:// handler
        # knows both cases.  Only the store is used -- no process group is created.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        store, _, _ = next(dist.rendezvous("env://", rank=rank, world_size=world_size))
        store = dist.PrefixStore("s3_comm", store)
        try:
            _comm, reason = RcclComm(rank, world_size, store), ""
        except Exception as err:         # librccl missing, communicator refused, ...
            _comm, reason = None, str(err)
        # every rank must end up with the same kind of communicator: agree through the store
        store.set(f"rccl_ok_{rank}", b"1" if _comm is not None else b"0")
        if all(store.get(f"rccl_ok_{r}") == b"1" for r in range(world_size)):
            _comm._store = store         # keep the rendezvous alive as long as the communicator
        else:
            import logging
            logging.getLogger(__name__).warning("RCCL communicator not available on every rank (%s): the exchange steps go "
                                                "through a gloo group instead.", reason or "another rank failed")
            if _comm is not None:
                _comm.close()
            own = not dist.is_initialized()
            if own:
                dist.init_process_group("gloo", rank=rank, world_size=world_size)
            _comm = GlooComm()
            _comm._own_group = own
            _comm.name = "gloo (RCCL communicator could not be created)"
    else:
        _comm = SoloComm()
    return _comm


def get_comm():
    """the communicator of this process: what ``init`` created, else a gloo group that torch.distributed already has,
    else a single-process stand-in"""
    global _comm
    if _comm is not None:
        return _comm
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if "gloo" in str(dist.get_backend()):
            return init("gloo")
        # an RCCL process group of the caller: bootstrap over its store
        _comm = RcclComm(dist.get_rank(), dist.get_world_size(), dist.distributed_c10d._get_default_store())
        return _comm
    return SoloComm()


def shutdown():
    global _comm
    if _comm is not None:
        _comm.close()
        if getattr(_comm, "_own_group", False):
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        _comm = None


def world():
    c = get_comm()
    return c.rank, c.world
