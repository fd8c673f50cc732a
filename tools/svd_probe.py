"""dev helper: time of the weighted Gram kernel (f64 MFMA) and of the whole weighted SVD at the bench grid's size"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
from sparsespatialsampling_amd import metrics, svd
n, t = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (461130, 1000)
x = pt.randn((n, t), dtype=pt.float64, device="cuda")
area = pt.rand(n, dtype=pt.float64, device="cuda") + 0.1
mean = metrics.temporal_mean(x)
g = svd.weighted_gram(x, mean, area); pt.cuda.synchronize()
e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    g = svd.weighted_gram(x, mean, area)
e1.record(); pt.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flops_alg = n * t * (t + 1)                     # upper triangle incl. diagonal: t (t + 1) / 2 entries x 2 flops x n rows
nb = -(-t // 128)
flops_issued = nb * (nb + 1) // 2 * 128 * 128 * 2 * n
print(f"weighted_gram N={n} T={t}: {ms:.2f} ms = {flops_alg / ms / 1e9:.1f} TFLOP/s algorithmic ({flops_alg / ms / 1e9 / 78.6:.3f} of the 78.6 TF f64 "
      f"matrix peak), {flops_issued / ms / 1e9:.1f} TFLOP/s issued; reads {n * t * 8 / 1e9:.2f} GB once = {n * t * 8 / ms / 1e6:.0f} GB/s")
t0 = time.perf_counter()
s, u, v = svd.compute_svd(x, area, rank=50)
pt.cuda.synchronize()
print(f"compute_svd rank 50: {time.perf_counter() - t0:.3f} s (mean + Gram + eigh + mode GEMM), s[0:3] = {s[:3].tolist()}")
t0 = time.perf_counter()
xw = (x - x.mean(-1, keepdim=True)) * area.sqrt()[:, None]
gref = xw.T @ xw
pt.cuda.synchronize()
print(f"torch (materialised centring + rocBLAS dgemm X^T X): {time.perf_counter() - t0:.3f} s, max rel diff {float((g - gref).abs().max() / gref.abs().max()):.2e}")
