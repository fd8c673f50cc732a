"""dev helper: host -> device upload strategies for one snapshot batch [N, T] fp32 into 128-B-pitched device rows"""
import sys, time
import torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import hipops
N = 4_991_774
for T in (25, 200):
    d = pt.randn((N, T), dtype=pt.float32)
    rows = hipops.padded_rows(N, T, pt.float32, "cuda")
    def t_(f, reps=3):
        f(); pt.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        pt.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    gb = d.numel() * 4 / 1e9
    a = t_(lambda: rows.copy_(d))
    print(f"T={T}: {gb:.2f} GB  direct strided copy_      {a*1e3:8.1f} ms  {gb/a:6.1f} GB/s", flush=True)
    dev = pt.empty((N, T), dtype=pt.float32, device="cuda")
    b = t_(lambda: (dev.copy_(d), rows.copy_(dev)))
    print(f"T={T}:          contiguous H2D + device repitch {b*1e3:8.1f} ms  {gb/b:6.1f} GB/s", flush=True)
    for slab_mb in (64, 256):
        slab = max(1, slab_mb * (1 << 20) // (T * 4))
        pins = [pt.empty((slab, T), dtype=pt.float32).pin_memory() for _ in range(2)]
        evs = [pt.cuda.Event() for _ in range(2)]
        def piped():
            i = 0
            for s0 in range(0, N, slab):
                s1 = min(N, s0 + slab)
                evs[i].synchronize()                       # the pinned buffer is free again
                pins[i][: s1 - s0].copy_(d[s0:s1])         # host memcpy (torch parallelises it)
                rows[s0:s1].copy_(pins[i][: s1 - s0], non_blocking=True)
                evs[i].record()
                i ^= 1
        c = t_(piped)
        print(f"T={T}:          pinned double buffer {slab_mb:4d} MB      {c*1e3:8.1f} ms  {gb/c:6.1f} GB/s", flush=True)
    dp = d.pin_memory()
    e = t_(lambda: rows.copy_(dp, non_blocking=True))
    print(f"T={T}:          source already pinned            {e*1e3:8.1f} ms  {gb/e:6.1f} GB/s", flush=True)
    del d, dp, rows, dev
