"""dev helper: randomised differential test of the persistent planned kernel (interp_planned_stream_kernel: k = 8 | 26, rows of
more than four 16-byte vectors) against the direct gather kernel (bit-equal results expected) and, on a slice, against
the oracle: row lengths 1 .. 600 incl. odd / ragged ones, f32 / f64, pitched and dense batches, partial tiles, plans with
fewer tiles than persistent workgroups.
    S3_STREAM_MIN_TILES=1 python tools/fuzz_stream.py [seed] [cases]"""
import os, sys
os.environ.setdefault("S3_STREAM_MIN_TILES", "1")
import numpy as np, torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import hipops
from oracle import s3_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
for case in range(n_cases):
    d = int(rng.integers(2, 4))
    k = int(rng.choice([8, 26]))
    n = int(rng.integers(k + 1, 80000))
    nc = int(rng.choice([1, 3, 63, 64, 65, int(rng.integers(1, 60000))]))
    f64 = bool(rng.random() < 0.3)
    epv = 2 if f64 else 4
    row_len = int(rng.choice([int(rng.integers(4 * epv + 1, 8 * epv + 1)), 25, 75, 100, int(rng.integers(1, 600))]))
    x = rng.random((n, d))
    c = rng.random((nc, d)) * (1.4 if rng.random() < 0.5 else 1.0) - 0.2
    knn = hipops.KnnIndex(x)
    idx, dist = knn.query(c, k)
    knn.close()
    w = hipops.idw_weights(dist)
    dtype = pt.float64 if f64 else pt.float32
    pad = bool(rng.random() < 0.6) or row_len % epv != 0
    if pad:
        data = hipops.padded_rows(n, row_len, dtype, "cuda", int(rng.integers(0, 3)))
    else:
        data = pt.empty((n, row_len), dtype=dtype, device="cuda")
    data.normal_()
    plan = hipops.InterpPlan(idx, n, c)
    got = plan.interp(w, data)
    ref = hipops.interp(w, idx, data.contiguous())
    ok = pt.equal(got, ref)
    why = "planned != direct"
    if ok and case % 8 == 0 and nc * k * row_len < 3e7:
        o = orc.interp(w.cpu().numpy(), idx.cpu().numpy(), data.contiguous().cpu().numpy().reshape(n, 1, row_len))
        g = got.cpu().numpy()
        ok = np.abs(o.reshape(nc, row_len) - g).max() <= 1e-13 * max(1.0, np.abs(g).max())
        why = "kernel != oracle (1e-13)"
    plan.close()
    if not ok:
        bad += 1
        print("MISMATCH", why, dict(case=case, d=d, n=n, k=k, nc=nc, f64=f64, row_len=row_len, pad=pad, tiles=plan.n_tiles), flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
