"""
Caller side of the export path under the reference's module name ``sparseSpatialSampling.utils`` (reference utils.py): the
batching loop that feeds ``ExportData.export`` (``export_openfoam_fields``, utils.py:155-226) and the weighted SVD helpers
(``compute_svd`` / ``write_svd_s_cube_to_file``, utils.py:302-413; implemented in ``svd.py`` on the f64 matrix cores).

The OpenFOAM readers themselves (``load_original_Foam_fields``, ``load_foam_data``: thin wrappers around flowtorch's
``FOAMDataloader``, utils.py:23-153, 228-299) are outside the hot path: they exist here under their
names so that ``from sparseSpatialSampling.utils import ...`` works, and raise ``ImportError`` WHEN CALLED.
``export_openfoam_fields`` therefore takes the reader as an argument (``loader=``): any callable with the signature of
``load_original_Foam_fields`` -- the reference's own function, or a reader of another solver's output.
"""
import logging
from typing import Callable, Union

from .svd import compute_svd, write_svd_s_cube_to_file  # noqa: F401  (re-exported under the reference's names)

logger = logging.getLogger(__name__)

_NO_FOAM = ("{name}() reads OpenFOAM cases through flowtorch's FOAMDataloader; that reader is not part of the MI355X build "
            "of the S^3 hot path.  Load the fields with flowtorch (or any other reader) yourself and call "
            "ExportData.export(coordinates, data, field_name, n_snapshots_total), or hand your reader to "
            "export_openfoam_fields(..., loader=your_function).")


def load_original_Foam_fields(load_dir: str, n_dimensions: int, boundaries: list, field_names: Union[list, str] = None,
                              write_times: Union[list, str] = None, get_field_names_and_times: bool = False):
    """signature of the reference's reader (utils.py:23-26); not available in this build -- see the module docstring"""
    raise ImportError(_NO_FOAM.format(name="load_original_Foam_fields"))


def load_foam_data(load_dir: str, boundaries: list, field_name="p", n_dims: int = 2, t_start: Union[int, float] = 0.4,
                   scalar: bool = True):
    """signature of the reference's reader (utils.py:228-229); not available in this build -- see the module docstring"""
    raise ImportError(_NO_FOAM.format(name="load_foam_data"))


def export_openfoam_fields(datawriter, load_path: str, boundaries: list, batch_size: int = None,
                           fields: Union[list, str] = None, loader: Callable = None) -> None:
    """Interpolate every requested field of a case onto the S^3 grid and export it, ``batch_size`` snapshots at a time
    (reference utils.py:155-226; what examples/s3_for_cylinder2D_Re100.py calls after the grid generation).

    :param datawriter: the ``ExportData`` object returned by the grid generation
    :param load_path: where the original CFD data lives (handed to ``loader`` untouched)
    :param boundaries: bounds of the numerical domain, as used for the grid generation
    :param batch_size: snapshots interpolated and written per ``export()`` call; ``None``: all at once
    :param fields: field name or list of names; ``None``: every field present at the first write time
    :param loader: reader with the signature of ``load_original_Foam_fields``:
        ``loader(path, n_dimensions, boundaries, get_field_names_and_times=True) -> (write_times, field_names)`` and
        ``loader(path, n_dimensions, boundaries, field_names=f, write_times=[...]) -> (coordinates, data)`` with ``data``
        ``[N, n_comp, T]``, or ``(None, None)`` for a field that does not exist.  Default: the flowtorch-backed reader of
        the reference, which this build does not ship (ImportError when called).
    """
    read = loader if loader is not None else load_original_Foam_fields
    n_dims = datawriter.n_dimensions
    # what the case holds is asked for only when the caller did not say (as the reference: one query per missing item)
    if fields is None:
        _, fields = read(load_path, n_dims, boundaries, get_field_names_and_times=True)
    if datawriter.write_times is None:
        datawriter.write_times, _ = read(load_path, n_dims, boundaries, get_field_names_and_times=True)
    times = datawriter.write_times
    n_total = len(times)
    per_batch = n_total if batch_size is None else int(batch_size)
    if per_batch < 1:
        raise ValueError(f"batch_size must be a positive number of snapshots, got {batch_size}")
    n_batches = -(-n_total // per_batch)
    for name in ([fields] if isinstance(fields, str) else list(fields)):
        for number, first in enumerate(range(0, n_total, per_batch), start=1):
            logger.info(f"Exporting batch {number} / {n_batches}")
            coordinates, data = read(load_path, n_dims, boundaries, field_names=name, write_times=times[first:first + per_batch])
            if data is None:                      # the reader found no such field for these times: nothing to export
                continue
            datawriter.export(coordinates, data, name, n_snapshots_total=n_total)
