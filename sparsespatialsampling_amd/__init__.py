"""
sparsespatialsampling_amd -- MI355X-native hot path of Sparse Spatial Sampling (S^3).

Drop-in for the grid-generation + snapshot-interpolation path of JanisGeise/sparseSpatialSampling: the public classes
keep the reference's names and signatures (``SparseSpatialSampling``, ``s_cube.SamplingTree``, ``export.ExportData``,
``data.Datawriter``, ``geometry.*``) while the numerical work runs in hand-written gfx950 HIP kernels
(``libs3hip.so``, C ABI in ``include/s3hip.h``).  There is no CPU compute path.
"""
from .version import __version__
