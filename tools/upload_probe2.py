"""dev helper: host -> device upload of FRESH pageable batches (as a user hands them over): direct copy_ vs a pipeline
through persistent pinned staging buffers filled by several host threads"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
import torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import hipops
N = 4_991_774
for T in (25, 200):
    rows = hipops.padded_rows(N, T, pt.float32, "cuda")
    gb = N * T * 4 / 1e9
    def fresh():
        return pt.randn((N, T), dtype=pt.float32)
    for rep in range(3):
        d = fresh()
        pt.cuda.synchronize(); t0 = time.perf_counter()
        rows.copy_(d); pt.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"T={T} direct copy_ of a fresh tensor: {dt*1e3:7.1f} ms {gb/dt:5.1f} GB/s", flush=True)
    for rep in range(4):
        d = fresh()
        pt.cuda.synchronize(); t0 = time.perf_counter()
        hipops.upload_rows(d, rows); pt.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"T={T} native staged upload (s3_upload_rows): {dt*1e3:7.1f} ms {gb/dt:5.1f} GB/s", flush=True)
    chk = fresh(); hipops.upload_rows(chk, rows); pt.cuda.synchronize(); assert pt.equal(rows.cpu(), chk)
    for n_thr, slab_mb in ((16, 128),):
        slab = max(1, slab_mb * (1 << 20) // (T * 4))
        pins = [pt.empty((slab, T), dtype=pt.float32).pin_memory() for _ in range(3)]
        evs = [pt.cuda.Event() for _ in range(3)]
        pool = ThreadPoolExecutor(n_thr)
        def fill(dst, src):
            n = src.shape[0]
            step = (n + n_thr - 1) // n_thr
            list(pool.map(lambda a: dst[a:min(n, a + step)].copy_(src[a:min(n, a + step)]), range(0, n, step)))
        for rep in range(3):
            d = fresh()
            pt.cuda.synchronize(); t0 = time.perf_counter()
            i = 0
            for s0 in range(0, N, slab):
                s1 = min(N, s0 + slab)
                evs[i].synchronize()
                fill(pins[i][: s1 - s0], d[s0:s1])
                rows[s0:s1].copy_(pins[i][: s1 - s0], non_blocking=True)
                evs[i].record()
                i = (i + 1) % 3
            pt.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"T={T} pinned pipeline {n_thr:2d} threads, {slab_mb:3d} MB slabs: {dt*1e3:7.1f} ms {gb/dt:5.1f} GB/s", flush=True)
        pool.shutdown()
        del pins
