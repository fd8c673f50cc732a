"""
Multi-GPU layer of the S^3 path: one process per GPU, the collectives run inside libs3hip.so on RCCL over xGMI
(``s3_comm_*``, csrc/comm.hip).  SURVEY.md 8(e):

* **interpolation** -- the generated cells are split into spatially compact, cost-balanced shards (``LeafShards``: stretches
  of the cells' Hilbert curve; the cut needs one small all-gather of cost profiles, no rank builds the table of all cells);
  every rank builds the KNN cache / plan of its shard, uploads only the source rows that shard references and computes its
  own output rows.  The bench step has no collective; ``ExportData``, which ends in ONE file, lets every rank write its rows
  -- transposed to snapshot-major on the way -- into ONE host buffer all ranks of the node map (``SharedHostArray``), each
  through its own PCIe link; the rank that writes the file hands that buffer to the HDF5 writer.  Nothing crosses xGMI.
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
    interpolation: stretches of the targets' Hilbert curve -- a stretch is a compact blob that shares few
    source rows with the other ranks -- of equal cost (bytes moved per snapshot: staged rows incl. halo + output rows).

    No rank builds the neighbour table of ALL targets (at 10^7 cells that table and its plan would be replicated on every
    GPU).  Every rank sorts the targets along the curve (keys + one radix sort: cheap), takes the r-th of ``world`` stretches
    of equal LENGTH, queries the neighbours of that stretch only and publishes its cumulative cost profile (``PROFILE``
    samples; one small all-gather of ``world * (PROFILE + 1)`` doubles); all ranks then cut the curve at equal COST from the
    same numbers.  A rank's final stretch differs from its first one by the imbalance only; ``ExportData`` queries it once
    more for the table it keeps.  Without a communicator (``comm=None``: tests, single-GPU probes) the profiles of all
    stretches are computed here, one after the other -- the same numbers, hence the same cuts.

    ``mine``      int64 host array, the targets of this rank (ascending ids)
    ``counts``    targets per rank; ``offsets``: first row of each rank's block in the gathered ``[n, L]`` array
    ``slot_of``   device int32 [n]: target id -> row of that array (the rank that collects the blocks restores the order)
    """
    PROFILE = 256

    def __init__(self, knn, targets, k, rank, world_size, comm=None):
        from . import hipops
        targets = hipops.to_device(targets, pt.float64)
        self.rank, self.world, self.n = int(rank), int(world_size), int(targets.shape[0])
        if self.n < self.world:
            raise ValueError(f"{self.n} target points cannot be sharded over {self.world} ranks")
        order = hipops.spatial_order(targets) if self.n > 1 else pt.zeros(1, dtype=pt.int32, device=targets.device)
        first = [shard_range(self.n, r, self.world)[0] for r in range(self.world)] + [self.n]      # equal lengths

        def profile_of(r):
            """cumulative cost along stretch r of the curve (cells in curve order: the plan keeps the order it is given)"""
            ids = order[first[r]:first[r + 1]].long()
            idx, _ = knn.query(targets[ids], k)
            plan = hipops.InterpPlan(idx, knn.n, None)
            prof = plan.cost_profile(self.PROFILE)
            plan.close()
            return prof

        profiles = pt.zeros((self.world, self.PROFILE + 1), dtype=pt.float64, device=targets.device)
        if comm is not None and comm.world == self.world and self.world > 1:
            profiles[self.rank] = pt.from_numpy(profile_of(self.rank)).to(targets.device)
            comm.allgather_inplace([profiles], [self.PROFILE + 1])
        else:
            for r in range(self.world):
                profiles[r] = pt.from_numpy(profile_of(r)).to(targets.device)
        cuts = self._cut(profiles.cpu().numpy(), first)
        self.counts = [cuts[r + 1] - cuts[r] for r in range(self.world)]
        self.offsets = [int(c) for c in cuts[:-1]]
        order_h = order.cpu().numpy()
        self.mine = np.sort(order_h[cuts[self.rank]:cuts[self.rank + 1]]).astype(np.int64)
        slot = np.empty(self.n, dtype=np.int32)
        for r in range(self.world):
            ids = np.sort(order_h[cuts[r]:cuts[r + 1]])
            slot[ids] = cuts[r] + np.arange(len(ids), dtype=np.int32)
        self.slot_of = pt.from_numpy(slot).to(order.device)

    def _cut(self, profiles, first):
        """positions along the curve that split the total cost into ``world`` equal parts (piecewise linear cumulative cost
        from the per-stretch profiles); every stretch keeps at least one target"""
        pos, cum, base = [], [], 0.0
        for r in range(self.world):
            length = first[r + 1] - first[r]
            pos.append(first[r] + length * np.arange(self.PROFILE + 1) / self.PROFILE)
            cum.append(base + profiles[r])
            base += profiles[r][-1]
        pos, cum = np.concatenate(pos), np.concatenate(cum)
        goals = base * np.arange(1, self.world) / self.world
        inner = np.rint(np.interp(goals, cum, pos)).astype(np.int64) if base > 0 else np.asarray(first[1:-1], dtype=np.int64)
        cuts = [0] + [int(c) for c in inner] + [self.n]
        for r in range(1, self.world):                       # strictly increasing, room for the ranks behind
            cuts[r] = min(max(cuts[r], cuts[r - 1] + 1), self.n - (self.world - r))
        return cuts


def batch_slice(n, rank, world_size):
    """equal-sized chunks for an in-place all-gather: (chunk, begin, end) -- rank r owns [r * chunk, (r + 1) * chunk)
    clipped to n; the gathered array spans world_size * chunk entries"""
    chunk = -(-int(n) // int(world_size))
    begin = min(rank * chunk, n)
    return chunk, begin, min(begin + chunk, n)


class SoloComm:
    rank, world, name = 0, 1, "solo"
    n_collectives = 0            # data-path collectives issued through this communicator (all-gathers, gathers to the root)

    def allgather_inplace(self, arrays, counts):
        pass

    def gather_to_root(self, send, recv, counts, root=0):
        """rank r's rows ``send`` [counts[r], L] -> rank ``root``'s ``recv`` [sum(counts), L], blocks in rank order; ``recv`` is
        None on the other ranks"""
        if recv is not None and send is not None and recv.data_ptr() != send.data_ptr():
            recv[:send.shape[0]].copy_(send)

    def allreduce_max(self, value):
        return float(value)

    def barrier(self):
        pass

    def broadcast_bytes(self, data, root=0):
        """``data`` of rank ``root`` on every rank (host-side plumbing: names of shared-memory segments)"""
        return data

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
        self.n_collectives += 1
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

    def gather_to_root(self, send, recv, counts, root=0):
        self.n_collectives += 1
        if self.rank != root:
            if counts[self.rank]:
                self._dist.send(send.cpu().contiguous(), dst=root)
            return
        off = 0
        for r, cnt in enumerate(counts):
            if cnt:
                if r == root:
                    recv[off:off + cnt].copy_(send)
                else:
                    host = pt.empty((cnt,) + tuple(recv.shape[1:]), dtype=recv.dtype)
                    self._dist.recv(host, src=r)
                    recv[off:off + cnt].copy_(host)
            off += cnt

    def allreduce_max(self, value):
        t = pt.tensor([float(value)], dtype=pt.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        self._dist.barrier()

    def broadcast_bytes(self, data, root=0):
        box = [data if self.rank == root else None]
        self._dist.broadcast_object_list(box, src=root)
        return box[0]


class RcclComm(SoloComm):
    """communicator inside libs3hip.so (RCCL); device tensors only"""
    name = "rccl"

    def __init__(self, rank, world_size, store):
        from . import _lib, hipops
        self._lib, self._ops = _lib, hipops
        self.rank, self.world = int(rank), int(world_size)
        lib = _lib.hip_lib()
        ident = (C.c_char * 128)()
        # ncclCommInitRank is a collective without a timeout: nobody may enter it unless everybody will.  Rank 0 publishes
        # the id OR an error marker (so that the others do not wait for an id that never comes); the callers of this
        # constructor have agreed beforehand that the library loads on every rank (parallel.init: preflight)
        if self.rank == 0:
            try:
                hipops.check(lib.s3_comm_unique_id(ident, 128), "s3_comm_unique_id")
            except Exception as err:
                store.set("s3_comm_id", b"ERR:" + str(err).encode()[:200])
                raise
            store.set("s3_comm_id", b"ID::" + bytes(ident.raw))
        else:
            blob = bytes(store.get("s3_comm_id"))
            if blob[:4] != b"ID::":
                raise RuntimeError(f"rank 0 could not create the RCCL id: {blob[4:].decode(errors='replace')}")
            ident.raw = blob[4:4 + 128]
        self._h = C.c_void_p(0)
        self._store, self._n_broadcasts = store, 0           # (the rendezvous store stays the host-side channel)
        hipops.device()
        hipops.check(lib.s3_comm_init(ident, 128, self.rank, self.world, C.byref(self._h)), "s3_comm_init")
        self._scalar = pt.zeros(1, dtype=pt.float64, device=hipops.device())

    def allgather_inplace(self, arrays, counts):
        """``arrays[i]`` is a contiguous device tensor whose first ``world * counts[i]`` elements are gathered in place:
        rank r contributes elements [r * counts[i], (r + 1) * counts[i]).  One RCCL group for all arrays."""
        n = len(arrays)
        self.n_collectives += 1
        ptrs = (C.c_void_p * n)(*[a.data_ptr() for a in arrays])
        sizes = (C.c_size_t * n)(*[int(c) * a.element_size() for a, c in zip(arrays, counts)])
        for a, c in zip(arrays, counts):
            if not (a.is_cuda and a.is_contiguous() and a.numel() >= self.world * c):
                raise ValueError("allgather_inplace: contiguous device tensors with room for world * count elements required")
        self._ops.check(self._lib.hip_lib().s3_comm_allgather_inplace(self._h, ptrs, sizes, n, self._ops._stream()),
                        "s3_comm_allgather_inplace")

    def gather_to_root(self, send, recv, counts, root=0):
        """device tensors: ``send`` [counts[rank], L] contiguous, ``recv`` [sum(counts), L] contiguous on ``root`` (None elsewhere);
        grouped ncclSend / ncclRecv inside the library -- every byte crosses xGMI once"""
        self.n_collectives += 1
        row = int(np.prod(send.shape[1:])) * send.element_size() if send is not None else int(np.prod(recv.shape[1:])) * recv.element_size()
        sizes = (C.c_size_t * self.world)(*[int(c) * row for c in counts])
        if send is not None and not (send.is_cuda and send.is_contiguous() and send.shape[0] == counts[self.rank]):
            raise ValueError("gather_to_root: contiguous device tensor with counts[rank] rows required")
        if self.rank == root and not (recv is not None and recv.is_cuda and recv.is_contiguous() and recv.shape[0] == sum(counts)):
            raise ValueError("gather_to_root: the root needs a contiguous device tensor with sum(counts) rows")
        self._ops.check(self._lib.hip_lib().s3_comm_gather_to_root(
            self._h, C.c_void_p(send.data_ptr() if send is not None and send.numel() else 0),
            C.c_void_p(recv.data_ptr() if recv is not None else 0), sizes, int(root), self._ops._stream()), "s3_comm_gather_to_root")

    def _allreduce(self, value, op):
        self._scalar[0] = float(value)
        self._ops.check(self._lib.hip_lib().s3_comm_allreduce_f64(self._h, C.c_void_p(self._scalar.data_ptr()), 1, op,
                                                                  self._ops._stream()), "s3_comm_allreduce_f64")
        return float(self._scalar.item())

    def allreduce_max(self, value):
        return self._allreduce(value, 1)

    def barrier(self):
        self._allreduce(0.0, 0)

    def broadcast_bytes(self, data, root=0):
        """through the rendezvous store the communicator was bootstrapped over (a blocking get on the other ranks)"""
        key = f"bcast_{self._n_broadcasts}"
        self._n_broadcasts += 1
        if self.rank == root:
            self._store.set(key, bytes(data))
            return bytes(data)
        return bytes(self._store.get(key))

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


class SharedMemoryUnavailable(OSError):
    """raised on EVERY rank when the node's shared-memory file system cannot hold a batch buffer (a container with a 64-MB /dev/shm)"""


def _boot_id():
    try:
        return open("/proc/sys/kernel/random/boot_id").read().strip()
    except OSError:
        return ""


class SharedHostArray:
    """``n_bytes`` of host memory that every rank of the node maps (POSIX shared memory) and that every rank's GPU can write
    through its own PCIe link (``s3_host_register``): the snapshot-major batch buffer of the sharded export.  Collective:
    every rank constructs it at the same point of the program.  The root creates ``/dev/shm/<name>``, the others open it by
    the broadcast name, and once everybody holds a mapping the root unlinks the name -- nothing is left behind whatever
    happens to the processes later.  ``array``: uint8 numpy view; ``device_ptr``: the mapping as this rank's kernels see it
    (None where registration is not possible: the caller then copies through a buffer of its own)."""
    _serial = 0

    def __init__(self, comm, n_bytes, register=True):
        import mmap
        import socket
        self.n_bytes = int(n_bytes)
        SharedHostArray._serial += 1
        proposal, fd = None, None
        if comm.rank == 0:
            # the root creates the segment BEFORE it tells anybody its name: a failure here (no room -- tmpfs pages are committed
            # when touched, a buffer that does not fit ends in SIGBUS, not in an error --, no permission, a name that exists) is
            # broadcast as an empty name, so that all ranks leave together instead of waiting at a barrier the root never reaches
            try:
                st = os.statvfs("/dev/shm")
                if st.f_bavail * st.f_frsize < self.n_bytes + (64 << 20):
                    raise OSError("not enough room in /dev/shm")
                name = f"s3_{os.getpid()}_{SharedHostArray._serial}"
                fd = os.open(os.path.join("/dev/shm", name), os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                os.ftruncate(fd, max(self.n_bytes, 1))
                # (the host's identity travels with the name: a rank on another node must not look for the segment in ITS /dev/shm)
                proposal = (name + "\n" + socket.gethostname() + "\n" + _boot_id()).encode()
            except OSError:
                if fd is not None:                           # created, but not usable (ftruncate failed): the name must not stay behind
                    os.close(fd)
                    fd = None
                    try:
                        os.unlink(os.path.join("/dev/shm", name))
                    except OSError:
                        pass
                proposal = b""
        msg = comm.broadcast_bytes(proposal).decode()
        if not msg:                                          # (decided by the root, learnt by everybody: all ranks take the same way)
            raise SharedMemoryUnavailable(f"/dev/shm cannot hold a batch buffer of {self.n_bytes} bytes")
        name, root_host, root_boot = msg.split("\n", 2)
        path = os.path.join("/dev/shm", name)
        ok, self._map = True, None
        try:
            if (socket.gethostname(), _boot_id()) != (root_host, root_boot):
                raise OSError("this rank runs on another host than rank 0")
            if comm.rank != 0:
                fd = os.open(path, os.O_RDWR)
            self._map = mmap.mmap(fd, max(self.n_bytes, 1))
        except OSError:
            ok = False
        finally:
            if fd is not None:
                os.close(fd)
        # everybody reports; nobody goes on with a buffer that somebody else does not have
        everybody = comm.allreduce_max(0.0 if ok else 1.0) == 0.0
        if comm.rank == 0:
            os.unlink(path)                                  # the mappings keep the pages; nothing is left behind whatever happens later
        if not everybody:
            if self._map is not None:
                self._map.close()
                self._map = None
            raise SharedMemoryUnavailable("not every rank could map the shared batch buffer (ranks on several hosts, or no "
                                          "access to /dev/shm)")
        self.array = np.frombuffer(self._map, dtype=np.uint8)
        self.device_ptr = None
        if register and self.n_bytes:
            from . import _lib
            d_ptr = C.c_void_p(0)
            if _lib.hip_lib().s3_host_register(C.c_void_p(self.array.ctypes.data), self.n_bytes, C.byref(d_ptr)) == 0:
                self.device_ptr = d_ptr.value

    def close(self):
        if getattr(self, "device_ptr", None):
            from . import _lib
            _lib.hip_lib().s3_host_unregister(C.c_void_p(self.array.ctypes.data))
            self.device_ptr = None
        self.array = None
        if getattr(self, "_map", None) is not None:
            try:
                self._map.close()
            except BufferError:                              # a tensor view is still alive: the mapping goes with it
                pass
            self._map = None

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
        # preflight: can every rank load the RCCL library and see its device?  Exchanged through the store BEFORE anybody
        # enters ncclCommInitRank, which would wait for a rank that never arrives
        try:
            from . import _lib, hipops
            hipops.device()
            able = bool(_lib.hip_lib().s3_comm_available())
            why = "" if able else "librccl could not be loaded"
        except Exception as err:
            able, why = False, str(err)
        store.set(f"can_rccl_{rank}", b"1" if able else b"0")
        everybody = all(store.get(f"can_rccl_{r}") == b"1" for r in range(world_size))
        _comm, reason = None, why or "another rank cannot use RCCL"
        if everybody:
            # ncclCommInitRank has no timeout and nothing inside a rank can see that ANOTHER rank never arrived: a timer ends this
            # process loudly if the bootstrap is still running after S3_COMM_INIT_TIMEOUT_S (300; 0 = wait forever), so that a
            # launcher (torch.distributed.run, bench.py's own parent) sees a failed rank instead of a silent hang and can end the rest
            import threading
            limit = float(os.environ.get("S3_COMM_INIT_TIMEOUT_S", "300"))

            def _overdue():
                import sys
                print(f"[s3] rank {rank}: the RCCL communicator bootstrap (ncclCommInitRank, {world_size} ranks) did not finish within "
                      f"{limit:.0f} s -- another rank is missing or wedged; ending this process (S3_COMM_INIT_TIMEOUT_S)", file=sys.stderr, flush=True)
                os._exit(86)
            timer = threading.Timer(limit, _overdue) if limit > 0 else None
            if timer is not None:
                timer.daemon = True
                timer.start()
            try:
                _comm, reason = RcclComm(rank, world_size, store), ""
            except Exception as err:     # id creation failed on rank 0 (every rank sees the marker), communicator refused
                _comm, reason = None, str(err)
            finally:
                if timer is not None:
                    timer.cancel()
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
        _comm = RcclComm(dist.get_rank(), dist.get_world_size(), dist.PrefixStore("s3_comm", dist.distributed_c10d._get_default_store()))
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
