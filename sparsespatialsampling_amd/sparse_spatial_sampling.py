"""
User-facing entry point of S^3 grid generation.

Drop-in for ``sparseSpatialSampling.sparse_spatial_sampling`` of the reference (class ``SparseSpatialSampling`` at
reference lines 20-186, ``list_geometries`` at 190-212): same constructor signature, same public attributes
(``coordinates, metric, save_path, save_name, grid_name, centers, vertices, faces, levels, n_dimensions,
size_initial_cell, n_jobs``), same side effects of ``execute_grid_generation`` (``mesh_info_<name>.pt`` and the pickled
``s_cube_<name>.pt``) and the same exception types for invalid input.  The work itself is done by
``s_cube.SamplingTree`` on the GPU; after grid generation the tree (and with it every device handle) is dropped, so the
object that gets pickled holds CPU tensors only.
"""
import inspect
import logging
import os
import textwrap
from typing import Union

import torch as pt

from .s_cube import SamplingTree

logger = logging.getLogger(__name__)


def _opt_int(value):
    return None if value is None else int(value)


class SparseSpatialSampling:
    def __init__(self, coordinates: pt.Tensor, metric: pt.Tensor, geometry_objects: list, save_path: str,
                 save_name: str, grid_name: str = "grid_s_cube", uniform_levels: int = 5,
                 n_cells_max: Union[int, float] = None, min_metric: float = 0.75, max_delta_level: bool = False,
                 n_cells_iter_start: int = None, n_cells_iter_end: int = None, n_jobs: int = 1,
                 relTol: Union[int, float] = 1e-3, reach_at_least: float = 0.75, pre_select_cells: bool = False):
        """Arguments, defaults and meaning as in the reference constructor (sparse_spatial_sampling.py:21-77):
        original cell centres ``[N, d]`` + per-cell metric ``[N]``, the geometry objects (one of them with
        ``keep_inside=True``), where to save, and the refinement controls handed on to ``SamplingTree``.  ``n_jobs`` is
        accepted for compatibility; the GPU path starts no worker processes."""
        # public state read by ExportData (reference export.py:74-83) and by user scripts
        self.n_jobs = n_jobs
        self.coordinates, self.metric = coordinates, metric
        self.save_path, self.save_name, self.grid_name = save_path, save_name, grid_name
        self.n_dimensions = coordinates.squeeze().size(-1)
        self.centers = self.vertices = self.faces = self.levels = None
        self.size_initial_cell = None

        self._geometries = geometry_objects
        self._pre_select_cells = pre_select_cells
        self._level_bounds = int(uniform_levels)
        self._n_cells_max = _opt_int(n_cells_max)
        self._min_metric = min_metric
        self._max_delta_level = max_delta_level
        self._n_cells_iter_start = _opt_int(n_cells_iter_start)
        self._n_cells_iter_end = _opt_int(n_cells_iter_end)
        self._relTol = relTol
        self._reach_at_least = reach_at_least

        self._check_input()
        self._sampling = SamplingTree(
            self.coordinates, self.metric, self._geometries,
            n_cells=self._n_cells_max, uniform_level=self._level_bounds, min_metric=self._min_metric,
            max_delta_level=self._max_delta_level, n_cells_iter_start=self._n_cells_iter_start,
            n_cells_iter_end=self._n_cells_iter_end, n_jobs=self.n_jobs, relTol=self._relTol,
            reach_at_least=self._reach_at_least, pre_select=self._pre_select_cells)

    def execute_grid_generation(self) -> None:
        """Run S^3, take over the grid and persist the mesh info and this object (reference lines 116-146)."""
        # with several ranks (one process per GPU, parallel.py) every rank generates the same grid and rank 0 writes the files
        from . import parallel
        writes = parallel.get_comm().rank == 0
        if writes:
            os.makedirs(self.save_path, exist_ok=True)
        tree = self._sampling
        tree.refine()
        if writes:
            pt.save(tree.data_final_mesh, os.path.join(self.save_path, f"mesh_info_{self.save_name}.pt"))

        self.centers, self.levels = tree.all_centers, tree.all_levels
        self.vertices, self.faces = tree.all_nodes, tree.face_ids
        self.size_initial_cell = tree.data_final_mesh["size_initial_cell"]

        # drop the tree: frees the device arrays and leaves a CPU-only, picklable object
        tree.close()
        self._sampling = None
        if writes:
            pt.save(self, os.path.join(self.save_path, f"s_cube_{self.save_name}.pt"))

    def _check_input(self) -> None:
        """Validate / repair the user input (same conditions and exception types as reference lines 148-186)."""
        shape = tuple(self.metric.size())
        assert len(shape) == 1, (f"The metric has to be one value per original cell, i.e. a 1D tensor of length "
                                 f"{self.coordinates.size(0)}; got a tensor of size {shape}.")
        assert self._geometries, "At least one geometry object (the numerical domain) is required, none was given."
        assert any(g.keep_inside for g in self._geometries), (
            "None of the geometry objects has 'keep_inside = True'. Exactly the numerical domain must be marked that way.")

        if self._n_cells_max is None and self._min_metric > 1:
            logger.warning("min_metric > 1 cannot be reached, using min_metric = 1 instead.")
            self._min_metric = 1
        if self._level_bounds <= 0:
            logger.warning(f"uniform_levels = {self._level_bounds} is invalid, using one uniform refinement cycle.")
            self._level_bounds = 1
        if self._n_cells_max is not None:
            logger.warning("'n_cells_max' is set: it replaces 'min_metric' as stopping criterion. Pass "
                           "'n_cells_max = None' to stop on the captured metric instead.")


def list_geometries() -> None:
    """Log every available geometry class with its one-line description (reference lines 190-212)."""
    from . import geometry
    base = geometry.GeometryObject
    found = {name: cls for name, cls in inspect.getmembers(geometry, inspect.isclass)
             if cls is not base and issubclass(cls, base)}
    width = max(map(len, found))
    lines = ["\n\tAvailable geometry objects:", "\t---------------------------"]
    for name in sorted(found):
        summary = textwrap.shorten(getattr(found[name], "__short_description__", ""), width=100, placeholder="…")
        lines.append(f"\t\t- {name.ljust(width)} : {summary}")
    lines.append("\n\tFor a more detailed description check out the documentation.")
    logger.info("\n".join(lines))
