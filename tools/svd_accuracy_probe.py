"""dev helper: which part limits the accuracy of the deflated spectrum (svd._spectrum) at the small end -- the Gram kernel, the
device eigen-solver or the residual GEMM: the same nine-decade test matrix with each of them swapped for the host version"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
from sparsespatialsampling_amd import svd, metrics, hipops
pt.set_default_dtype(pt.float64)
rng = np.random.default_rng(5)
n, t = 6000, 40
q1, _ = np.linalg.qr(rng.standard_normal((n, t)))
q2, _ = np.linalg.qr(rng.standard_normal((t, t)))
area = rng.random(n) * 0.5 + 0.05
centred = (q1 * 10.0 ** np.linspace(0, -9, t)) @ q2.T
centred -= centred.mean(1, keepdims=True)
data = pt.from_numpy(centred / np.sqrt(area)[:, None] + rng.standard_normal((n, 1)) * 5.0)
w = pt.from_numpy(area)
xw = (data - data.mean(-1, keepdim=True)) * w.sqrt()[:, None]
s_ref = pt.linalg.svdvals(xw)
x2, wd = data.cuda(), w.cuda()
mean = metrics.temporal_mean(x2)
real_gram, real_eigh = svd.weighted_gram, svd._eigh
def host_gram(x, mu, ww):
    a = ((x - mu.reshape(-1, 1)) * ww.sqrt().reshape(-1, 1)).cpu()
    return (a.T @ a).cuda()
def host_eigh(g):
    return pt.linalg.eigh(g.cpu())
def scaled_eigh(g):
    scale = g.diagonal().max()
    lam, vec = pt.linalg.eigh(g / scale)
    return (lam * scale).cpu(), vec.cpu()
for name, gram, eigh in (("device gram + scaled device eigh", real_gram, scaled_eigh), ("device gram + device eigh", real_gram, real_eigh), ("host gram + device eigh", host_gram, real_eigh),
                         ("device gram + host eigh", real_gram, host_eigh), ("host gram + host eigh", host_gram, host_eigh)):
    svd.weighted_gram, svd._eigh = gram, eigh
    s, _ = svd._spectrum(x2, mean, wd, None)
    err = (s.cpu() - s_ref).abs()
    print(f"{name:28s} max abs error {float(err.max()):.3e}; last six {err[-6:].numpy()}", flush=True)
