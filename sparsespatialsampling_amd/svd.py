"""
Weighted SVD of the fields on the S^3 grid -- the consumer downstream of the interpolation.
Mirrors ``sparseSpatialSampling.utils.compute_svd`` (reference utils.py:302-346): temporal mean removed, every cell
weighted by the square root of its area / volume (``Dataloader.weights``, data.py:240-247), returns ``(s, U, V)`` with the
weighting divided out of the modes again.

How it runs on the MI355X (method of snapshots; the snapshot count T is small against the number of cells N):

1. temporal mean per row -- ``s3_row_moments`` (one streaming pass, csrc/metric.hip);
2. ``G = sum_n a_n (x_n - mean_n)(x_n - mean_n)^T`` [T, T] -- ``s3_weighted_gram`` on the f64 matrix cores
   (``v_mfma_f64_16x16x4_f64``, csrc/svd.hip); centring and weighting are fused into the operand staging, the data matrix
   is read as the interpolation kernel left it (f64, HBM resident);
3. ``G = V diag(s^2) V^T`` -- symmetric eigenproblem of a T x T matrix: the vendor's dense solver, the ONE library call of the path
   (rocSOLVER's ``dsyevd`` behind the C ABI entry ``s3_sym_eig``, csrc/eig.hip -- looked up with dlopen, no torch operator involved;
   ``bench.py --workload svd`` reports it as ``"eigh": "library"`` with its share of a
   ``compute_svd`` call.  A hand-written solver was priced and not built: one-sided Jacobi needs ~10 sweeps x (T - 1) rounds with a
   chip-wide synchronisation each -- 10^4 x ~5 us at T = 1000, more than the 23 ms of the library's tridiagonal solver -- and the
   blocked form leaves 16 workgroups busy);
4. modes ``U = (X - mean) V diag(1/s)`` -- ``s3_centered_gemm`` on the same matrix cores, the centring fused into the operand
   staging; the weights cancel: ``(sqrt(a) (X - mean) V / s) / sqrt(a)``.

The reference delegates to ``flowtorch.analysis.SVD`` (absent here): ``rank=None`` selects the optimal hard threshold of
Gavish & Donoho as flowtorch documents it (``opt_rank``); that selection rule is restated from the documentation, not pinned
against flowtorch.  Unlike the reference the caller's ``data_matrix`` is not modified.

Accuracy.  An eigenvalue of the Gram matrix carries an absolute error of about eps * s_max^2, i.e. a singular value s a
relative error of eps * (s_max / s)^2: fine down to s / s_max ~ 1e-3 (1e-10), useless below 1e-8.  The leading modes -- what
``write_svd_s_cube_to_file`` stores -- never get there, but the optimal-rank rule takes the MEDIAN of the spectrum, which for
low-noise data lies far below.  Whenever a singular value that matters (all of them for ``rank=None``, the requested ones
otherwise) falls under 1e-3 * s_max the spectrum is therefore refined by DEFLATION (``_spectrum``): the modes found so far
are projected out of the data, the Gram matrix of the residual -- whose largest singular value is now 1e-3 of the previous
level's -- is taken by the same kernel, and so on for up to three levels; the result has the absolute accuracy of a direct
SVD (about eps * s_max; ``torch.linalg.svd`` of the float64 matrix is the checker,
tests/test_gpu_kernels.py::test_compute_svd_small_singular_values).  "Matters" is read narrowly: the optimal-rank
rule needs the spectrum down to its MEDIAN only, and the direction the centring removed (eigenvalue ~0 by construction) is
never waited for -- a full-rank noisy matrix takes ONE Gram pass.  The residual is never materialised as a second N x T
matrix: the coefficients ``A = (X - mean) V`` [N, found] are kept, and ``(X - mean) - A V^T`` is formed in row chunks of
``RESIDUAL_ROWS`` that go straight into the Gram kernel (both by ``s3_centered_gemm``).
"""
import ctypes as C
from typing import Tuple

import torch as pt

from . import _lib, hipops, metrics


def optimal_rank(s: pt.Tensor, n_rows: int, n_cols: int) -> int:
    """singular values above the optimal hard threshold ``omega(beta) * median(s)`` (Gavish & Donoho 2014, unknown noise
    level; ``beta`` = aspect ratio <= 1)"""
    beta = min(n_rows, n_cols) / max(n_rows, n_cols)
    omega = 0.56 * beta ** 3 - 0.95 * beta ** 2 + 1.82 * beta + 1.43
    tau = omega * float(pt.median(s))
    return max(1, int((s > tau).sum()))


def _eigh(g: pt.Tensor):
    """symmetric eigenproblem of the small T x T Gram matrix through the C ABI (``s3_sym_eig``: rocSOLVER's dense solver looked up by
    libs3hip.so itself, scaled to a unit diagonal maximum around the call -- csrc/eig.hip; 1000 x 1000 float64: ~0.1 s against ~0.8 s
    with LAPACK on the host).  Returns (eigenvalues ascending, eigenvectors in COLUMNS) on the host, as ``torch.linalg.eigh`` does.
    A process in which rocSOLVER cannot be loaded takes LAPACK on the host (``torch.linalg.eigh`` of the downloaded matrix)."""
    lib = _lib.hip_lib()
    t = int(g.shape[0])
    if not (g.is_cuda and g.dtype == pt.float64 and g.is_contiguous()):
        raise TypeError("_eigh: contiguous float64 device matrix required")
    if not lib.s3_sym_eig_available():
        scale = g.diagonal().max()
        lam, vec = pt.linalg.eigh((g / scale).cpu() if bool(scale > 0) else g.cpu())
        return (lam * scale.cpu() if bool(scale > 0) else lam), vec
    lam = pt.empty(t, dtype=pt.float64, device=g.device)
    rows = pt.empty((t, t), dtype=pt.float64, device=g.device)                   # eigenvector j in ROW j
    scratch = pt.empty(int(lib.s3_sym_eig_scratch_bytes(t)), dtype=pt.uint8, device=g.device)
    hipops.check(lib.s3_sym_eig(hipops._ptr(g), t, hipops._ptr(lam), hipops._ptr(rows), hipops._ptr(scratch), hipops._stream()), "s3_sym_eig")
    return pt.from_numpy(hipops.to_host(lam)), pt.from_numpy(hipops.to_host(rows)).T


LEVEL_RANGE = 1e-3          # singular values down to this fraction of a level's largest one are taken from that level's Gram matrix
MAX_LEVELS = 4
RESIDUAL_ROWS = 1 << 16     # rows of the residual that exist at a time (a chunk is formed, enters the Gram matrix and is dropped)


def centered_gemm(left: pt.Tensor, left_mean, b: pt.Tensor, minus_from: pt.Tensor = None, minus_from_mean=None) -> pt.Tensor:
    """``(left - left_mean 1^T) @ b`` or, with ``minus_from``, ``(minus_from - minus_from_mean 1^T) - (left - left_mean 1^T) @ b``
    on the f64 matrix cores (s3_centered_gemm).  ``left`` [m, k] and ``minus_from`` [m, n] may be row-pitched device matrices
    (unit inner stride), ``b`` [k, n]; means are device vectors [m] or None.  Returns a contiguous [m, n] device matrix."""
    for name, a in (("left", left), ("b", b)) + ((("minus_from", minus_from),) if minus_from is not None else ()):
        if not (a.is_cuda and a.dtype == pt.float64 and a.dim() == 2 and a.stride(1) == 1):
            raise TypeError(f"centered_gemm: {name} has to be a 2-D float64 device matrix with unit inner stride")
    m, k, n = int(left.shape[0]), int(left.shape[1]), int(b.shape[1])
    b = b.contiguous()
    if int(b.shape[0]) != k or (minus_from is not None and tuple(minus_from.shape) != (m, n)):
        raise ValueError("centered_gemm: shapes do not match")
    out = pt.empty((m, n), dtype=pt.float64, device=left.device)
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out.zero_() if minus_from is None else out.copy_(minus_from if minus_from_mean is None else minus_from - minus_from_mean.reshape(-1, 1))
    ptr = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
    hipops.check(_lib.hip_lib().s3_centered_gemm(ptr(left), m, k, int(left.stride(0)), ptr(left_mean), ptr(b), n, ptr(minus_from),
                                                 int(minus_from.stride(0)) if minus_from is not None else 0, ptr(minus_from_mean),
                                                 ptr(out), hipops._stream()), "s3_centered_gemm")
    return out


def _residual_gram(x2: pt.Tensor, mean: pt.Tensor, w: pt.Tensor, basis: pt.Tensor) -> pt.Tensor:
    """Gram matrix of ``sqrt(w) * ((x2 - mean 1^T) (I - basis basis^T))`` without a second N x T matrix: the coefficients
    ``A = (x2 - mean) basis`` [N, found], then per chunk of rows the residual ``(x2 - mean) - A basis^T`` -> Gram kernel"""
    n, t = int(x2.shape[0]), int(x2.shape[1])
    coeff = centered_gemm(x2, mean, basis)                                        # [N, found]
    basis_t = basis.T.contiguous()
    gram = pt.zeros((t, t), dtype=pt.float64, device=x2.device)
    zero_mean = pt.zeros(min(n, RESIDUAL_ROWS), dtype=pt.float64, device=x2.device)
    for r0 in range(0, n, RESIDUAL_ROWS):
        r1 = min(n, r0 + RESIDUAL_ROWS)
        residual = centered_gemm(coeff[r0:r1], None, basis_t, minus_from=x2[r0:r1], minus_from_mean=mean[r0:r1])
        gram += weighted_gram(residual, zero_mean[:r1 - r0], w[r0:r1])
    return gram


def _spectrum(x2: pt.Tensor, mean: pt.Tensor, w: pt.Tensor, wanted: int):
    """all singular values (descending, host) and right singular vectors [T, T] (host) of sqrt(w) * (x2 - mean): Gram matrix +
    symmetric eigenproblem, refined by deflation where the first ``wanted`` values reach below ``LEVEL_RANGE`` of the largest
    one of a level (module docstring).  The last direction -- what the centring removed, eigenvalue ~0 -- is never waited for."""
    t = int(x2.shape[1])
    gram = weighted_gram(x2, mean, w)
    s_parts, v_parts, found = [], [], 0
    for level in range(MAX_LEVELS):
        if found:
            # The residual's Gram matrix has the directions already found as a null space, and the solver is free to mix them with
            # the small directions still wanted.  Instead of projecting the vectors afterwards (round 4: two T x T products and a QR
            # on the host) the found directions are SHIFTED out of the way first: G + sigma B B^T gives them the eigenvalue sigma, twice
            # the largest one of the residual -- an eigenvalue of its own, so the solver returns every other vector orthogonal to B
            # to rounding, and orthonormal among themselves as always.  B B^T on the f64 matrix cores (s3_centered_gemm).
            basis = pt.cat(v_parts, dim=1)                                        # [T, found], host
            # sigma: the residual's largest eigenvalue is at most the LAST ACCEPTED one (the spectrum is descending), so twice that
            # is an eigenvalue of its own.  (Round 5 took 2 ||G||_F, up to 2 sqrt(T) times larger: the solver's absolute tolerance
            # scales with sigma once the matrix is brought to a unit diagonal -- ADVICE r5.)
            sigma = 2.0 * float(s_parts[-1][-1]) ** 2
            if sigma > 0:
                # G + sigma B B^T = G - (-sigma B) B^T: one s3_centered_gemm with the Gram matrix as its "minus_from" operand
                gram = centered_gemm(hipops.to_device(-sigma * basis), None, hipops.to_device(basis.T.contiguous()), minus_from=gram)
        lam, vec = _eigh(gram)                                    # ascending; T x T
        lam, vec = lam.flip(0).clamp_min(0.0), vec.flip(1)
        lam, vec = lam[found:], vec[:, found:]                    # (the `found` directions come out first, at sigma: dropped)
        s_level = lam.sqrt()
        good = int((lam >= lam[0] * LEVEL_RANGE ** 2).sum()) if float(lam[0]) > 0 else 0
        meaningful = len(lam) - 1                                 # the centring's null direction never passes the test above
        need_more = good < meaningful and found + good < wanted and level + 1 < MAX_LEVELS and good > 0
        if not need_more:
            s_parts.append(s_level)
            v_parts.append(vec)
            break
        s_parts.append(s_level[:good])
        v_parts.append(vec[:, :good])
        found += good
        gram = _residual_gram(x2, mean, w, pt.cat(v_parts, dim=1).to(x2.device))
    return pt.cat(s_parts), pt.cat(v_parts, dim=1)


def weighted_gram(x: pt.Tensor, mean: pt.Tensor, weight: pt.Tensor) -> pt.Tensor:
    """``sum_n weight[n] (x[n] - mean[n]) (x[n] - mean[n])^T`` for a device matrix ``x`` [N, T] float64 (rows may be pitched:
    a column slice of a wider buffer) -> [T, T] float64 on the device"""
    if not (x.is_cuda and x.dtype == pt.float64 and x.dim() == 2 and x.stride(1) == 1):
        raise TypeError("weighted_gram: 2-D float64 device matrix with unit inner stride required")
    n, t = int(x.shape[0]), int(x.shape[1])
    mean, weight = hipops.to_device(mean, pt.float64).reshape(-1), hipops.to_device(weight, pt.float64).reshape(-1)
    if mean.numel() != n or weight.numel() != n:
        raise ValueError("weighted_gram: one mean and one weight per row required")
    lib = _lib.hip_lib()
    scratch = pt.empty(int(lib.s3_weighted_gram_scratch_bytes(n, t)), dtype=pt.uint8, device=x.device)
    g = pt.empty((t, t), dtype=pt.float64, device=x.device)
    hipops.check(lib.s3_weighted_gram(C.c_void_p(x.data_ptr()), n, t, int(x.stride(0)), hipops._ptr(mean), hipops._ptr(weight),
                                      hipops._ptr(g), hipops._ptr(scratch), hipops._stream()), "s3_weighted_gram")
    return g


def compute_svd(data_matrix: pt.Tensor, cell_area: pt.Tensor, rank: int = None) -> Tuple[pt.Tensor, pt.Tensor, pt.Tensor]:
    """weighted SVD of a field: ``data_matrix`` [N_cells, N_snapshots] (scalar) or [N_cells, N_dims, N_snapshots]
    (vector; the components are stacked like the reference does, utils.py:333-336), ``cell_area`` [N_cells].  Returns
    ``(s [r], U [N_cells, (N_dims,) r], V [N_snapshots, r])`` on the device the data came from."""
    shape = tuple(data_matrix.shape)
    if len(shape) not in (2, 3):
        raise ValueError(f"expected [N_cells, N_snapshots] or [N_cells, N_dims, N_snapshots], got {shape}")
    on_host = not data_matrix.is_cuda
    x = hipops.to_device(data_matrix if data_matrix.dtype == pt.float64 else data_matrix.to(pt.float64))
    area = hipops.to_device(cell_area, pt.float64).reshape(-1)
    n_cells, t = shape[0], shape[-1]
    if len(shape) == 3:
        # the reference reshapes [N, C, T] -> [C * N, T] of the C-contiguous tensor, i.e. row (n, c) keeps the area of cell n
        x2 = x.reshape(n_cells * shape[1], t)
        w = area.repeat_interleave(shape[1])
    else:
        x2, w = x, area
    mean = metrics.temporal_mean(x2)
    # rank=None: the optimal-rank rule compares with the MEDIAN of the spectrum -- values below it need not be accurate
    s_all, vec = _spectrum(x2, mean, w, t // 2 + 1 if rank is None else min(int(rank), t))
    r = optimal_rank(s_all, x2.shape[0], t) if rank is None else min(int(rank), t)
    keep = s_all[:r] > s_all[0] * 1e-14 if r else s_all[:0] > 0
    r = int(keep.sum()) if r else 0
    s, v = s_all[:r], vec[:, :r].contiguous()
    b = (v / s).to(x2.device)                                 # [T, r]
    u = centered_gemm(x2, mean, b)                            # (X - mean 1^T) B; the sqrt(area) factors cancel
    if len(shape) == 3:
        u = u.reshape(n_cells, shape[1], r)
    s, v = s.to(x2.device), v.to(x2.device)
    return (s.cpu(), u.cpu(), v.cpu()) if on_host else (s, u, v)


def write_svd_s_cube_to_file(field_names, load_dir: str, file_name: str, new_file: bool, n_modes: int = None, rank=None,
                             t_start=0) -> None:
    """weighted SVD of exported fields, written next to them as ``<file_name>_<field>_svd.h5`` + XDMF (signature and file
    contents of the reference's ``utils.write_svd_s_cube_to_file``, utils.py:349-413): the grid, ``constant/mode_<i>`` for
    the first ``n_modes`` modes, ``constant/V``, ``constant/s`` and ``constant/cell_area``.  Snapshots with a write time
    >= ``t_start`` are used, in ascending time."""
    import logging
    from .data import Dataloader, Datawriter
    log = logging.getLogger(__name__)
    for name in ([field_names] if isinstance(field_names, str) else list(field_names)):
        log.info(f"Performing SVD for field {name}.")
        loader = Dataloader(load_dir, f"{file_name}_{name}.h5" if new_file else f"{file_name}.h5", dtype=pt.float64)
        times = sorted((t for t in loader.write_times if float(t) >= t_start), key=float)
        s, u, v = compute_svd(loader.load_snapshot(name, times), loader.weights, rank)
        available = int(u.shape[-1])
        count = available if n_modes is None else n_modes
        if count > available:
            log.warning(f"Number of modes to write is set to {count}, but found only {available} modes to write.")
            count = available
        writer = Datawriter(load_dir, f"{file_name}_{name}_svd.h5")
        writer.write_grid(loader)
        for i in range(count):
            writer.write_data(f"mode_{i + 1}", group="constant", data=u[..., i])
        for key, values in (("V", v), ("s", s), ("cell_area", loader.weights)):
            writer.write_data(key, group="constant", data=values)
        writer.write_xdmf_file()
