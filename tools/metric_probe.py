"""dev helper: throughput of the temporal-moments kernel (s3_row_moments) on bench-sized snapshot matrices"""
import os, sys, statistics, ctypes as C
import torch as pt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsespatialsampling_amd import _lib, hipops
hipops.device()
for n, t, dtype in ((4_991_774, 1000, pt.float32), (4_991_774, 256, pt.float32), (4_991_774, 25, pt.float32), (2_000_000, 1000, pt.float64)):
    data = pt.empty((n, t), dtype=dtype, device="cuda").normal_()
    mean = pt.empty(n, dtype=pt.float64, device="cuda"); std = pt.empty_like(mean)
    def run():
        hipops.check(_lib.hip_lib().s3_row_moments(C.c_void_p(data.data_ptr()), hipops.DTYPE_CODE[dtype], n, t, t, 1,
                                                   C.c_void_p(mean.data_ptr()), C.c_void_p(std.data_ptr()), hipops._stream()), "m")
    run(); pt.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); pt.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    ms = statistics.median(ts)
    gb = n * t * data.element_size() / 1e9 + n * 16 / 1e9
    a, b = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    a.record(); ref = data.std(-1); b.record(); pt.cuda.synchronize()
    print(f"N={n} T={t} {dtype}: {ms:.3f} ms = {gb/ms*1e3:.0f} GB/s ({gb/ms/8:.3f} of 8 TB/s)   torch.std on the device: {a.elapsed_time(b):.2f} ms", flush=True)
    del data, ref
