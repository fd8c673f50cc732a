"""
Metric fields upstream of S^3: the temporal statistics the reference's example scripts compute with torch on the CPU
before they hand a metric to ``SparseSpatialSampling`` (``metric = pt.std(field, dim=1)``,
examples/s3_for_OAT15_airfoil.py:91).  One streaming pass over the snapshot matrix on the GPU
(``s3_row_moments``, csrc/metric.hip): f64 accumulation, results equal to ``torch.std`` / ``torch.mean`` of the float64
data to rounding (1e-12 relative).
"""
import ctypes as C

import numpy as np
import torch as pt

from . import _lib, hipops


def temporal_moments(field: pt.Tensor, unbiased: bool = True):
    """mean and standard deviation over the last axis of ``field`` ([N, T] or [N, n_comp, T], float32 / float64, host or
    device) -> (mean, std) float64 tensors of shape ``field.shape[:-1]`` on the device the field came from"""
    if field.dim() < 2:
        raise ValueError(f"expected a field of shape [N, T] or [N, n_comp, T], got {tuple(field.shape)}")
    if field.dtype not in hipops.DTYPE_CODE:
        field = field.to(pt.float64)
    on_host = not field.is_cuda
    dev = hipops.to_device(field)
    t = int(dev.shape[-1])
    n_rows = int(np.prod(dev.shape[:-1]))
    mean = pt.empty(n_rows, dtype=pt.float64, device=dev.device)
    std = pt.empty(n_rows, dtype=pt.float64, device=dev.device)
    hipops.check(_lib.hip_lib().s3_row_moments(C.c_void_p(dev.data_ptr()), hipops.DTYPE_CODE[dev.dtype], n_rows, t, t,
                                               1 if unbiased else 0, C.c_void_p(mean.data_ptr()),
                                               C.c_void_p(std.data_ptr()), hipops._stream()), "s3_row_moments")
    mean, std = mean.reshape(dev.shape[:-1]), std.reshape(dev.shape[:-1])
    if on_host:
        hipops.synchronize()
        return mean.cpu(), std.cpu()
    return mean, std


def temporal_mean_abs_sum(field: pt.Tensor) -> pt.Tensor:
    """``torch.mean(field.abs().sum(1), 1)`` of a field ``[N, n_comp, T]`` (the metric of the reference's cylinder2D script,
    examples/s3_for_cylinder2D_Re100.py:55) in one pass on the GPU, accumulated in float64: the sum of |x| over a cell's
    ``n_comp * T`` values divided by T"""
    if field.dim() != 3:
        raise ValueError(f"expected a field of shape [N, n_comp, T], got {tuple(field.shape)}")
    if field.dtype not in hipops.DTYPE_CODE:
        field = field.to(pt.float64)
    on_host = not field.is_cuda
    dev = hipops.to_device(field)
    n, n_comp, t = (int(v) for v in dev.shape)
    mean = pt.empty(n, dtype=pt.float64, device=dev.device)
    hipops.check(_lib.hip_lib().s3_row_abs_moments(C.c_void_p(dev.data_ptr()), hipops.DTYPE_CODE[dev.dtype], n, n_comp * t,
                                                   n_comp * t, 0, C.c_void_p(mean.data_ptr()), None, hipops._stream()),
                 "s3_row_abs_moments")
    mean *= float(n_comp)                                   # mean over n_comp * T values -> sum over components, mean over T
    if on_host:
        hipops.synchronize()
        return mean.cpu()
    return mean


def temporal_std(field: pt.Tensor, unbiased: bool = True) -> pt.Tensor:
    """``torch.std(field, dim=-1)`` (unbiased by default, as torch) computed in float64 in one pass on the GPU"""
    return temporal_moments(field, unbiased)[1]


def temporal_mean(field: pt.Tensor) -> pt.Tensor:
    """``torch.mean(field, dim=-1)`` computed in float64 on the GPU"""
    return temporal_moments(field)[0]


class RunningMoments:
    """mean / standard deviation over time of a field that arrives in snapshot BATCHES (the reference's cylinder3D script never
    holds all snapshots at once, examples/s3_for_cylinder3D_Re3900.py:28-69): every batch is reduced by one streaming pass on the
    GPU (``temporal_moments``) and merged into the running (count, mean, M2) per cell with the pairwise update of Chan et al. --
    float64 throughout, equal to ``torch.std`` / ``torch.mean`` over the concatenated snapshots to rounding"""

    def __init__(self):
        self.count, self._mean, self._m2 = 0, None, None

    def update(self, batch: pt.Tensor) -> "RunningMoments":
        n_b = int(batch.shape[-1])
        if n_b == 0:
            return self
        mean_b, std_b = temporal_moments(batch, unbiased=False)
        m2_b = std_b * std_b * n_b
        if self.count == 0:
            self.count, self._mean, self._m2 = n_b, mean_b, m2_b
            return self
        n = self.count + n_b
        delta = mean_b - self._mean
        self._mean = self._mean + delta * (n_b / n)
        self._m2 = self._m2 + m2_b + delta * delta * (self.count * n_b / n)
        self.count = n
        return self

    def mean(self) -> pt.Tensor:
        return self._mean

    def std(self, unbiased: bool = True) -> pt.Tensor:
        return (self._m2 / (self.count - 1 if unbiased else self.count)).sqrt()


def tke_from_uprime2mean(prime2mean: pt.Tensor) -> pt.Tensor:
    """turbulent kinetic energy ``0.5 * (u'u' + v'v' + w'w')`` from one snapshot of OpenFOAM's ``UPrime2Mean`` field -- the
    metric of the reference's cylinder3D script, examples/s3_for_cylinder3D_Re3900.py:104
    (``0.5 * prime2Mean[:, [0, 3, 5]].sum(-1)``; a ``volSymmTensorField`` stores [XX, XY, XZ, YY, YZ, ZZ]).  No time axis to
    reduce: three loads and two additions per cell, evaluated where the tensor lives (host or device), in the tensor's own
    precision, with the reference's operation order."""
    if prime2mean.dim() != 2 or prime2mean.shape[1] != 6:
        raise ValueError(f"expected a symmetric-tensor field [N, 6] (XX, XY, XZ, YY, YZ, ZZ), got {tuple(prime2mean.shape)}")
    return 0.5 * prime2mean[:, [0, 3, 5]].sum(-1)
