"""dev helper: what the PCIe link of the box gives for the export path's transfers -- pinned D2H / H2D as one copy, in
pieces on several streams, both directions at once, and the snapshot-major transpose kernel writing STRAIGHT into pinned
host memory (no device copy of the transposed batch, no separate download).
    python tools/transport_probe.py [GiB]"""
import sys, time, ctypes as C
import numpy as np, torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import _lib, hipops

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(gib * 2 ** 30 / 8)
nc, t = 461_130, max(1, n // 461_130)
n = nc * t
dev = pt.empty(n, dtype=pt.float64, device="cuda").normal_()
host = pt.empty(n, dtype=pt.float64, pin_memory=True)
host2 = pt.empty(n, dtype=pt.float64, pin_memory=True).normal_()
dev2 = pt.empty(n, dtype=pt.float64, device="cuda")
pt.cuda.synchronize()


def timed(fn, reps=3):
    fn(); pt.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    pt.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


gb = n * 8 / 1e9
dt = timed(lambda: host.copy_(dev, non_blocking=True))
print(f"D2H one copy, pinned, {gb:.2f} GB: {dt * 1e3:.1f} ms = {gb / dt:.1f} GB/s", flush=True)
dt = timed(lambda: dev2.copy_(host2, non_blocking=True))
print(f"H2D one copy, pinned: {dt * 1e3:.1f} ms = {gb / dt:.1f} GB/s", flush=True)
s1, s2 = pt.cuda.Stream(), pt.cuda.Stream()


def duplex():
    with pt.cuda.stream(s1):
        host.copy_(dev, non_blocking=True)
    with pt.cuda.stream(s2):
        dev2.copy_(host2, non_blocking=True)


dt = timed(duplex)
print(f"D2H + H2D at once (two streams): {dt * 1e3:.1f} ms = {gb / dt:.1f} GB/s each way", flush=True)
for pieces in (4, 16):
    streams = [pt.cuda.Stream() for _ in range(min(pieces, 4))]
    step = n // pieces

    def split():
        for i in range(pieces):
            with pt.cuda.stream(streams[i % len(streams)]):
                host[i * step:(i + 1) * step].copy_(dev[i * step:(i + 1) * step], non_blocking=True)
    dt = timed(split)
    print(f"D2H in {pieces} pieces on {len(streams)} streams: {dt * 1e3:.1f} ms = {gb / dt:.1f} GB/s", flush=True)

# the transpose kernel with its output in pinned host memory (the device reaches it through the same address)
lib = _lib.hip_lib()
vals = dev.view(nc, t)


def fused():
    hipops.check(lib.s3_snapshot_major(hipops._ptr(vals), nc, 1, t, C.c_void_p(host.data_ptr()), hipops._stream()), "s3_snapshot_major")


try:
    dt = timed(fused)
    ok = bool(pt.equal(host.view(t, nc), vals.t().contiguous().cpu()))
    print(f"snapshot_major straight into pinned host memory: {dt * 1e3:.1f} ms = {gb / dt:.1f} GB/s, correct {ok}", flush=True)
except Exception as err:
    print("snapshot_major into host memory failed:", err)
dt = timed(lambda: host.copy_(hipops.snapshot_major(vals, 1, t).view(-1), non_blocking=True))
print(f"snapshot_major on the device + one D2H copy: {dt * 1e3:.1f} ms = {gb / dt:.1f} GB/s", flush=True)
