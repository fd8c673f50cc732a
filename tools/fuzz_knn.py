"""dev helper: randomised differential test of the bucket-grid KNN (s3_knn_query / s3_idw_predict) against the oracle's
brute-force search: degenerate clouds (duplicates, collinear / coplanar points, clusters, lattices with exact ties),
queries far outside the cloud, k up to 64"""
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
    n = int(rng.integers(1, 6000))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        x = rng.random((n, d))
    elif kind == 1:                                      # clusters over decades
        x = rng.random((n, d)) * 10.0 ** rng.integers(-6, 1, (n, 1))
    elif kind == 2:                                      # lattice: many exactly equidistant neighbours
        x = rng.integers(0, 9, (n, d)).astype(np.float64) / 8.0
    elif kind == 3:                                      # flat cloud (one coordinate constant)
        x = rng.random((n, d)); x[:, rng.integers(0, d)] = 0.375
    elif kind == 4:                                      # duplicates
        x = rng.random((max(1, n // 7), d))[rng.integers(0, max(1, n // 7), n)]
    else:                                                # a line
        x = np.outer(rng.random(n), rng.random(d)) + 0.1
    k = int(rng.integers(1, min(64, n) + 1))
    nq = int(rng.integers(1, 3000))
    q = rng.random((nq, d)) * 10.0 ** rng.integers(-1, 2) - rng.random() * 2
    if rng.random() < 0.3:
        q[: nq // 2] = x[rng.integers(0, n, nq // 2)]    # exact hits
    y = rng.standard_normal(n)
    knn = hipops.KnnIndex(x, target_occupancy=float(rng.choice([0.0, 1.0, 4.0, 30.0])))
    knn.set_values(y)
    idx, dist = knn.query(q, k)
    pred = knn.predict(q, k)
    knn.close()
    oi, od = orc.knn(x, q, k)
    op = orc.idw_predict(x, y, q, k)
    ok = np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(dist.cpu().numpy(), od) and \
        np.array_equal(pred.cpu().numpy(), op)
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=case, d=d, n=n, kind=kind, k=k, nq=nq,
                               idx=bool(np.array_equal(idx.cpu().numpy(), oi)), dist=bool(np.array_equal(dist.cpu().numpy(), od)),
                               pred=bool(np.array_equal(pred.cpu().numpy(), op))), flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
