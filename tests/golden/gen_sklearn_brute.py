"""Generator of tests/golden/sklearn_brute.npz: what scikit-learn returns where it switches to BRUTE FORCE -- `algorithm="auto"`
takes the kd-tree unless k >= N // 2 (sklearn/neighbors/_base.py:625-633), which only toy clouds reach (the reference's 50-point
unit tests).  There the squared distances come from the expansion |q|^2 + |p|^2 - 2 q.p through a BLAS product, so they differ from
the kd-tree's sum of squared differences in the last bits.  This fixture pins how far: the neighbour SETS and their order are those
of the exact search, distances and `weights="distance"` predictions agree to ~1e-14 relative.  scikit-learn itself is the source
here (installed in the dev container; it is a third-party dependency of the reference, not the reference).
    python tests/golden/gen_sklearn_brute.py"""
import os
import numpy as np
import sklearn
from sklearn.neighbors import KNeighborsRegressor, NearestNeighbors

rng = np.random.default_rng(2024)
out = {"sklearn_version": np.array(sklearn.__version__)}
cases = [(2, 50, 26), (2, 17, 8), (2, 12, 8), (3, 50, 26), (3, 30, 26), (3, 17, 8), (3, 53, 26), (2, 16, 8)]
for i, (d, n, k) in enumerate(cases):
    x, y, q = rng.random((n, d)), rng.random(n) + 0.1, rng.random((60, d)) * 1.2 - 0.1
    nb = NearestNeighbors(n_neighbors=k).fit(x)
    assert nb._fit_method == ("brute" if k >= n // 2 else "kd_tree")
    dist, idx = nb.kneighbors(q)
    pred = KNeighborsRegressor(n_neighbors=k, weights="distance").fit(x, y).predict(q)
    out.update({f"x{i}": x, f"y{i}": y, f"q{i}": q, f"idx{i}": idx.astype(np.int64), f"dist{i}": dist, f"pred{i}": pred,
                f"method{i}": np.array(nb._fit_method), f"k{i}": np.array(k)})
out["n_cases"] = np.array(len(cases))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sklearn_brute.npz"), **out)
print("wrote", len(cases), "cases; methods:", [str(out[f"method{i}"]) for i in range(len(cases))])
