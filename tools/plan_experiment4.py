"""dev helper (offline, no GPU): staged source rows of the interpolation plan for tiles that are 3-D blocks instead of runs of
the cells' Hilbert curve -- on the dumped neighbour table of the bench grid (tools/dump_plan_inputs.py:
gpurun_out/plan_inputs_cylinder3D_Re3900.npz).  Caps as in the plan builder: <= 64 cells and <= 496 distinct rows per tile.
    python tools/plan_experiment4.py [npz]"""
import sys
import numpy as np

z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/plan_inputs_cylinder3D_Re3900.npz")
centers, idx = z["centers"], z["idx"]
nc, k = idx.shape
TC, UCAP = 64, 496
print(f"{nc} cells, k = {k}, distinct rows referenced: {len(np.unique(idx))}")


def rows_of(cells):
    return len(np.unique(idx[cells]))


def hilbert_keys(c, bits=16):
    lo, hi = c.min(0), c.max(0)
    q = np.minimum(((c - lo) / (hi - lo).max() * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            sel = (X[i] & Q) != 0
            X[0] = np.where(sel, X[0] ^ P, X[0])
            t = np.where(sel, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    h = np.zeros(len(c), dtype=np.uint64)
    for bit in range(bits - 1, -1, -1):
        for i in range(3):
            h = (h << np.uint64(1)) | ((X[i] >> bit) & 1).astype(np.uint64)
    return h


def greedy_runs(order):
    """the plan builder's packing: consecutive cells of `order` while both caps hold"""
    tiles, staged, cur, rows = 0, 0, [], set()
    for c in order:
        new = rows | set(idx[c].tolist())
        if len(cur) + 1 > TC or len(new) > UCAP:
            tiles += 1
            staged += len(rows)
            cur, rows = [c], set(idx[c].tolist())
        else:
            cur.append(c)
            rows = new
    tiles += 1
    staged += len(rows)
    return tiles, staged


def bisect(cells, out, target_rows):
    """recursive coordinate bisection: boxes of cells, cut across their longest extent in proportion to the number of tiles each
    side needs, until a box holds <= 64 cells and <= 496 distinct rows"""
    n = len(cells)
    r = rows_of(cells) if n <= 4 * TC else UCAP + 1
    if n <= TC and r <= UCAP:
        out.append((n, r))
        return
    # how many tiles this box needs at least (by cells and, where known, by rows)
    parts = max(2, -(-n // TC), int(np.ceil(r / target_rows)) if n <= 4 * TC else 2)
    left = parts // 2
    c = centers[cells]
    axis = int(np.argmax(c.max(0) - c.min(0)))
    o = np.argsort(c[:, axis], kind="stable")
    cut = int(round(n * left / parts))
    cut = min(max(cut, 1), n - 1)
    bisect(cells[o[:cut]], out, target_rows)
    bisect(cells[o[cut:]], out, target_rows)


h = hilbert_keys(centers)
order = np.argsort(h, kind="stable")
t, s = greedy_runs(order)
print(f"runs of the Hilbert curve (the plan builder): {t} tiles, {s} staged rows ({s / len(np.unique(idx)):.3f} x distinct)")
base = s
for target in (496, 440, 400):
    out = []
    bisect(np.arange(nc), out, target)
    n_t, staged = len(out), sum(r for _, r in out)
    print(f"coordinate bisection into boxes (row target {target}): {n_t} tiles, {staged} staged rows = {staged / base:.3f} of the curve's, "
          f"mean {np.mean([c for c, _ in out]):.1f} cells / {np.mean([r for _, r in out]):.0f} rows per tile")
