"""dev helper: randomised differential test of the planned vs the direct interpolation kernel (bit-equal results
expected) over shapes, k, dtypes, row pitches and neighbour-table structures; a slice is also checked against the oracle"""
import sys
import numpy as np, torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import hipops
from oracle import s3_oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for case in range(n_cases):
    d = int(rng.integers(2, 4))
    n = int(rng.integers(30, 60000))
    k = int(rng.integers(1, min(64, n) + 1))
    nc = int(rng.integers(1, 20000))
    f64 = bool(rng.random() < 0.3)
    epv = 2 if f64 else 4
    ncomp = int(rng.integers(1, 4))
    t = int(rng.integers(1, 70))
    row_len = ncomp * t                                  # any length: ragged rows live in a padded pitch
    if rng.random() < 0.25:
        row_len = int(rng.integers(1, 700))              # also rows long enough for the chunk-pipelined kernel
    structure = rng.integers(0, 3)
    x = rng.random((n, d))
    if structure == 0:                                   # real neighbour table
        c = rng.random((nc, d)) * (1.4 if rng.random() < 0.5 else 1.0) - 0.2
        knn = hipops.KnnIndex(x)
        idx, dist = knn.query(c, k)
        knn.close()
        w = hipops.idw_weights(dist)
        centers = c
    else:                                                # arbitrary table (repeats inside a row allowed), optional centres
        idx = pt.from_numpy(rng.integers(0, n, (nc, k)).astype(np.int32)).cuda()
        if structure == 2:
            idx[:, k // 2:] = idx[:, : k - k // 2]       # duplicated neighbours
        w = pt.from_numpy(rng.random((nc, k))).cuda()
        centers = rng.random((nc, d)) if rng.random() < 0.5 else None
    dtype = pt.float64 if f64 else pt.float32
    pad = bool(rng.random() < 0.6) or row_len % epv != 0
    if pad:
        data = hipops.padded_rows(n, row_len, dtype, "cuda", int(rng.integers(0, 3)))
    else:
        data = pt.empty((n, row_len), dtype=dtype, device="cuda")
    data.normal_()
    plan = hipops.InterpPlan(idx, n, centers, tile_cells=int(rng.choice([0, 64, 128])))
    got = plan.interp(w, data)
    ref = hipops.interp(w, idx, data.contiguous())
    ok = pt.equal(got, ref)
    why = "planned != direct"
    if ok and case % 10 == 0 and nc * k * row_len < 3e7:
        o = orc.interp(w.cpu().numpy(), idx.cpu().numpy(), data.contiguous().cpu().numpy().reshape(n, 1, row_len))
        g = got.cpu().numpy()
        ok = np.abs(o.reshape(nc, row_len) - g).max() <= 1e-13 * max(1.0, np.abs(g).max())
        why = "kernel != oracle (1e-13)"
    plan.close()
    if not ok:
        bad += 1
        print("MISMATCH", why, dict(case=case, d=d, n=n, k=k, nc=nc, f64=f64, row_len=row_len, structure=int(structure), pad=pad), flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
