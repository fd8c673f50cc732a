"""
sparsespatialsampling_amd -- MI355X-native hot path of Sparse Spatial Sampling (S^3).

Drop-in for the grid-generation + snapshot-interpolation path of JanisGeise/sparseSpatialSampling: the public classes
keep the reference's names and signatures (``SparseSpatialSampling``, ``s_cube.SamplingTree``, ``export.ExportData``,
``data.Datawriter``, ``geometry.*``) while the numerical work runs in hand-written gfx950 HIP kernels
(``libs3hip.so``, C ABI in ``include/s3hip.h``).  There is no CPU compute path.
"""
import os as _os

# A hazard of the HIP runtime this package met in its own test processes (round 5; DESIGN §8, INTEGRATION "Deviations /
# hazards"): a copy between the device and PAGEABLE host memory of a megabyte or more is served by pinning the caller's pages on the
# fly, and rare processes ended with "Memory access fault by GPU ... write access to a read-only page" at a host heap address inside
# torch's ``tensor.cpu()`` right after this library's multi-threaded uploads.  The library itself stages every pageable copy through
# its own page-locked buffers; for the COPIES OF THE SAME PROCESS THAT TORCH MAKES (a user's ``.cpu()`` / ``.cuda()`` on pageable
# tensors) the runtime is told to use ITS staging buffers below 4 GiB.  It reads the variable when HIP is initialised, so this has to
# happen on import, before the first device call; S3_KEEP_RUNTIME_PINNING=1 leaves the runtime's default alone.
if _os.environ.get("S3_KEEP_RUNTIME_PINNING") != "1":
    _os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4096")

from .version import __version__
