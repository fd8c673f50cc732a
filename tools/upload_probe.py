"""dev helper: where the time of a small host -> device upload goes (hipops.to_device on a 20 KB id list)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
from sparsespatialsampling_amd import hipops, _lib

x = np.arange(5000, dtype=np.int32)
hipops.to_device(x); pt.cuda.synchronize()


def t(label, fn, n=200, sync_each=False):
    pt.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each:
            pt.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    pt.cuda.synchronize()
    print(f"{label:50s} {dt * 1e6:8.1f} us")


xt = pt.from_numpy(x)
dev = hipops.device()
out = pt.empty(5000, dtype=pt.int32, device=dev)
lib = _lib.hip_lib()
t("hipops.device()", lambda: hipops.device())
t("pt.from_numpy + ascontiguousarray", lambda: pt.from_numpy(np.ascontiguousarray(x)))
t("pt.empty on device", lambda: pt.empty(5000, dtype=pt.int32, device=dev))
t("hipops._stream()", lambda: hipops._stream())
t("s3_upload_rows 20 KB", lambda: lib.s3_upload_rows(C.c_void_p(xt.data_ptr()), 1, 20000, C.c_void_p(out.data_ptr()), 20000, hipops._stream()))
t("s3_upload_rows 20 KB, sync each", lambda: lib.s3_upload_rows(C.c_void_p(xt.data_ptr()), 1, 20000, C.c_void_p(out.data_ptr()), 20000, hipops._stream()), sync_each=True)
t("x.to(dev) 20 KB", lambda: xt.to(dev))
t("hipops.to_device 20 KB", lambda: hipops.to_device(x))
big = np.arange(250000, dtype=np.int32)
bt = pt.from_numpy(big)
t("hipops.to_device 1 MB", lambda: hipops.to_device(big))
t("x.to(dev) 1 MB", lambda: bt.to(dev))
flags = pt.zeros(40000, dtype=pt.uint8, device=dev)
t("flags.cpu() 40 KB", lambda: flags.cpu())
t("sync only", lambda: pt.cuda.synchronize())
