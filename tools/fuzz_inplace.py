"""dev helper: randomised differential test of tables READ IN PLACE (s3_interp_planned_src / s3_interp_planned on dense, pitched,
offset and strided tables): rows off the 128-byte grid through interp_planned_shift_kernel, element-aligned rows through the
persistent kernel, short rows -- against the direct gather kernel, bit for bit, and (every tenth case) against the oracle
    python tools/fuzz_inplace.py [seed] [cases]"""
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
    n = int(rng.integers(2000, 90000))
    k = int(rng.choice([8, 26, 26, 26, 5, 40]))
    nc = int(rng.integers(100, 30000))
    f64 = bool(rng.random() < 0.3)
    item = 8 if f64 else 4
    dtype = pt.float64 if f64 else pt.float32
    kind = rng.choice(["long16", "long16", "long", "mid", "short"])
    if kind == "long16":                                 # whole 16-byte vectors, more than eight chunks: the shift kernel
        row_len = int(rng.integers(33, 160)) * (16 // item) * int(rng.integers(1, 4))
    elif kind == "long":
        row_len = int(rng.integers(300, 1300))
    elif kind == "mid":
        row_len = int(rng.integers(17, 300))
    else:
        row_len = int(rng.integers(16 // item, 40))
    extra = int(rng.choice([0, 0, 16 // item, 3 * (16 // item), 1, 7]))          # pitch = row_len + extra elements
    lead = int(rng.choice([0, 0, 16 // item, 5 * (16 // item), 1]))              # the table starts `lead` elements into its buffer
    buf = pt.empty(lead + n * (row_len + extra), dtype=dtype, device="cuda").normal_()
    table = buf[lead:].view(n, row_len + extra)[:, :row_len]
    x = rng.random((n, d))
    c = rng.random((nc, d)) * (0.5 if rng.random() < 0.5 else 1.0) + 0.2
    knn = hipops.KnnIndex(x)
    idx, dist = knn.query(c, k)
    knn.close()
    w = hipops.idw_weights(dist)
    ref = hipops.interp(w, idx, table.contiguous())
    used, remap = hipops.referenced_rows([idx], n, coords=x if rng.random() < 0.5 else None)
    idx_c = idx.clone()
    hipops.remap_indices(idx_c, remap)
    plan = hipops.InterpPlan(idx_c, int(used.numel()), c)
    plan.set_weights(w)
    plan.set_source_ids(used.contiguous(), n)
    ok, why = True, ""
    if hipops.InterpPlan.supports(k, table):
        got = plan.interp_src(table)
        ok, why = bool(pt.equal(got, ref)), "in place != direct"
        if ok and case % 10 == 0 and nc * k * row_len < 3e7:
            o = orc.interp(w.cpu().numpy(), idx.cpu().numpy(), table.contiguous().cpu().numpy().reshape(n, 1, row_len))
            g = got.cpu().numpy()
            ok, why = bool(np.abs(o.reshape(nc, row_len) - g).max() <= 1e-13 * max(1.0, np.abs(g).max())), "kernel != oracle (1e-13)"
    else:
        why = "skipped"
    plan.close()
    if not ok:
        bad += 1
        print("MISMATCH", why, dict(case=case, d=d, n=n, k=k, nc=nc, f64=f64, row_len=row_len, extra=extra, lead=lead, kind=str(kind)), flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
