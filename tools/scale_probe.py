"""dev helper: wall-clock of refine + KNN cache + planned interpolation at other scales / dimensions (run from the repo root)"""
import sys, time, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)


def run(name, x, metric, geos, k, t, **kw):
    pt.cuda.synchronize(); t0 = time.perf_counter()
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw)
    tree.refine(); pt.cuda.synchronize(); t1 = time.perf_counter()
    centers = tree.all_centers.numpy(); info = tree.data_final_mesh
    ncells_total = tree._topo.n_cells
    tree._backend.close(); del tree
    knn = hipops.KnnIndex(x); idx, dist = knn.query(centers, k); w = hipops.idw_weights(dist); pt.cuda.synchronize(); t2 = time.perf_counter()
    plan = hipops.InterpPlan(idx, len(x), centers); pt.cuda.synchronize(); t3 = time.perf_counter()
    data = hipops.padded_rows(len(x), t, pt.float32, "cuda"); data.normal_()
    out = pt.empty((len(centers), t), dtype=pt.float64, device="cuda")
    plan.interp(w, data, out=out); pt.cuda.synchronize()
    e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): plan.interp(w, data, out=out)
    e1.record(); pt.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nu = int(pt.unique(idx).numel())
    balg = nu * t * 4 + len(centers) * t * 8 + len(centers) * k * 12
    print(f"{name}: N={len(x)} -> {len(centers)} leaves ({ncells_total} cells, {info['iterations']} it, levels {info['min_level']}-{info['max_level']}, "
          f"metric {info['metric_per_iter'][-1]:.3f}) refine {t1-t0:.2f}s [uni {info['t_uniform']:.2f} ada {info['t_adaptive']:.2f} geo {info['t_geometry']}] "
          f"knn cache {t2-t1:.3f}s plan {t3-t2:.3f}s interp T={t}: {ms:.3f} ms = {len(centers)*t/ms/1e6:.1f} G/s, alg {balg/ms/1e6:.0f} GB/s "
          f"staged/unique rows {plan.total_rows/nu:.2f}", flush=True)


which = sys.argv[1:] or ["c1", "c2", "big3d"]
if "c1" in which:    # cylinder2D-like
    rng = np.random.default_rng(0)
    x = rng.random((14500, 2)) * [2.2, 0.41]
    x = x[((x - [0.2, 0.2]) ** 2).sum(1) > 0.05 ** 2]
    m = 0.02 + np.exp(-((x[:, 1] - 0.2) / 0.08) ** 2) * np.where(x[:, 0] > 0.2, np.exp(-(x[:, 0] - 0.2)), 0) + np.exp(-20 * np.hypot(x[:, 0] - 0.2, x[:, 1] - 0.2))
    geos = [geometry.CubeGeometry("domain", True, [0, 0], [2.2, 0.41]), geometry.SphereGeometry("cylinder", False, [0.2, 0.2], 0.05, refine=True, min_refinement_level=9)]
    run("C1 cylinder2D", x, m, geos, 8, 400, uniform_level=5, min_metric=0.75)
if "c2" in which:    # OAT15-like: clustered points around a polygon
    rng = np.random.default_rng(1)
    sys.path.insert(0, "tests")
    n = 120; xs = 0.5 * (1 - np.cos(np.linspace(0, np.pi, n // 2))); yt = 0.6 * (0.2969 * np.sqrt(xs) - 0.126 * xs - 0.3516 * xs ** 2 + 0.2843 * xs ** 3 - 0.1036 * xs ** 4)
    poly = np.concatenate([np.stack([xs, yt], 1), np.stack([xs[::-1], -yt[::-1]], 1)[1:-1]])
    x = np.concatenate([rng.random((150000, 2)) * [1.4, 1.0] + [-0.2, -0.5], poly[rng.integers(0, len(poly), 150000)] + 0.02 * rng.standard_normal((150000, 2))])
    m = 0.05 + np.exp(-8 * np.abs(x[:, 1])) * (1 + np.sin(6 * x[:, 0]) ** 2)
    geos = [geometry.CubeGeometry("domain", True, [-0.2, -0.5], [1.2, 0.5]), geometry.GeometryCoordinates2D("airfoil", False, poly, refine=True)]
    run("C2 OAT15-like", x, m, geos, 8, 2000, uniform_level=5, n_cells=25000)
if "big3d" in which:  # 2e7 points (towards SURVEY 8(d) C4)
    cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"]); cfg["n"] = 20_000_000
    x, m = bench.synthetic_cylinder3d(cfg)
    geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
            geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
    run("big3d 2e7", x, m, geos, 26, 256, uniform_level=5, min_metric=0.75)
if "c4" in which:     # 5e7 points in the unit box, n_cells_max = 1e7, tiles of 16 snapshots
    rng = np.random.default_rng(3)
    n = 50_000_000
    x = rng.random((n, 3))
    r = np.sqrt(((x - 0.5) ** 2).sum(1))
    m = 0.05 + np.exp(-6 * r) * (1 + 0.5 * np.sin(25 * x[:, 0]) * np.cos(17 * x[:, 1]))
    del r
    geos = [geometry.CubeGeometry("domain", True, [0, 0, 0], [1, 1, 1])]
    run("C4 stress 5e7", x, m, geos, 26, 16, uniform_level=5, n_cells=10_000_000)
