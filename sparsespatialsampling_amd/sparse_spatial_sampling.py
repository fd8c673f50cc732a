"""
User-facing entry point of S^3 grid generation -- drop-in for the reference's
``sparseSpatialSampling/sparse_spatial_sampling.py`` (``SparseSpatialSampling`` :20-186, ``list_geometries`` :190-212).

The object stays picklable after ``execute_grid_generation()`` (reference :146 saves it with ``torch.save`` and the
examples reload it): the GPU-backed ``SamplingTree`` is dropped before saving and only CPU tensors remain.
"""
import inspect
import logging
import textwrap
from os import makedirs, path
from os.path import join
from typing import Union

import torch as pt

from .s_cube import SamplingTree

logger = logging.getLogger(__name__)


class SparseSpatialSampling:
    def __init__(self, coordinates: pt.Tensor, metric: pt.Tensor, geometry_objects: list, save_path: str,
                 save_name: str, grid_name: str = "grid_s_cube", uniform_levels: int = 5,
                 n_cells_max: Union[int, float] = None, min_metric: float = 0.75, max_delta_level: bool = False,
                 n_cells_iter_start: int = None, n_cells_iter_end: int = None, n_jobs: int = 1,
                 relTol: Union[int, float] = 1e-3, reach_at_least: float = 0.75, pre_select_cells: bool = False):
        """Arguments as in the reference (sparse_spatial_sampling.py:21-77)."""
        self.n_jobs = n_jobs
        self.coordinates = coordinates
        self.metric = metric
        self.save_path = save_path
        self.save_name = save_name
        self.grid_name = grid_name
        self.centers = None
        self.vertices = None
        self.faces = None
        self.n_dimensions = coordinates.squeeze().size(-1)
        self.size_initial_cell = None
        self.levels = None

        self._geometries = geometry_objects
        self._pre_select_cells = pre_select_cells
        self._level_bounds = int(uniform_levels)
        self._n_cells_max = n_cells_max if n_cells_max is None else int(n_cells_max)
        self._min_metric = min_metric
        self._max_delta_level = max_delta_level
        self._n_cells_iter_start = n_cells_iter_start if n_cells_iter_start is None else int(n_cells_iter_start)
        self._n_cells_iter_end = n_cells_iter_end if n_cells_iter_end is None else int(n_cells_iter_end)
        self._relTol = relTol
        self._reach_at_least = reach_at_least

        self._check_input()
        self._sampling = SamplingTree(self.coordinates, self.metric, self._geometries, n_cells=self._n_cells_max,
                                      uniform_level=self._level_bounds, min_metric=self._min_metric,
                                      max_delta_level=self._max_delta_level, n_cells_iter_end=self._n_cells_iter_end,
                                      n_cells_iter_start=self._n_cells_iter_start, n_jobs=self.n_jobs,
                                      relTol=self._relTol, reach_at_least=self._reach_at_least,
                                      pre_select=self._pre_select_cells)

    def execute_grid_generation(self) -> None:
        """run S^3, keep the grid, persist ``mesh_info_<name>.pt`` and ``s_cube_<name>.pt`` (reference :116-146)"""
        if not path.exists(self.save_path):
            makedirs(self.save_path)
        self._sampling.refine()
        pt.save(self._sampling.data_final_mesh, join(self.save_path, f"mesh_info_{self.save_name}.pt"))
        self.levels = self._sampling.all_levels
        self.centers = self._sampling.all_centers
        self.vertices = self._sampling.all_nodes
        self.faces = self._sampling.face_ids
        self.size_initial_cell = self._sampling.data_final_mesh["size_initial_cell"]
        self._sampling = None          # releases the device arrays; what is left is CPU-only and picklable
        pt.save(self, join(self.save_path, f"s_cube_{self.save_name}.pt"))

    def _check_input(self) -> None:
        assert len(self.metric.size()) == 1, (f"The size of the metric must be a 1D tensor of the length "
                                              f"{self.coordinates.size(0)}. The size of the metric given is "
                                              f"{self.metric.size()}.")
        if self._n_cells_max is None:
            if self._min_metric > 1:
                logger.warning("A value of min_metric > 1 is invalid. Changed min_metric to 1.")
                self._min_metric = self._min_metric if self._min_metric < 1 else 1
        assert self._geometries, ("No geometries are provided. Please provide at least one geometry for the "
                                  "numerical domain.")
        assert any([g.keep_inside for g in self._geometries]), ("No geometry for the domain provided. At least one "
                                                                "geometry object must have 'keep_inside = True' "
                                                                "representing the numerical domain.")
        if self._level_bounds <= 0:
            logger.warning(f"Lower level bound of {self._level_bounds} is invalid. Changed lower level bound to 1.")
            self._level_bounds = 1
        if self._n_cells_max is not None:
            logger.warning("Detected stopping criterion 'n_cells_max'. Passing this stopping criterion deactivates the"
                           " 'min_metric' stopping criterion. To use 'min_metric' as stopping criterion, remove "
                           "'n_cells_max' or set 'n_cells_max = None'.")


def list_geometries() -> None:
    """log the available geometry classes with their one-line description (reference :190-212)"""
    from . import geometry
    from .geometry.geometry_base import GeometryObject
    classes = sorted((obj for _, obj in inspect.getmembers(geometry, inspect.isclass)
                      if issubclass(obj, GeometryObject) and obj is not GeometryObject), key=lambda c: c.__name__)
    pad = max(len(cls.__name__) for cls in classes)
    msg = ["\n\tAvailable geometry objects:", "\t---------------------------"]
    for cls in classes:
        desc = textwrap.shorten(getattr(cls, "__short_description__", ""), width=100, placeholder="…")
        msg.append(f"\t\t- {cls.__name__.ljust(pad)} : {desc}")
    msg.append("\n\tFor a more detailed description check out the documentation.")
    logger.info("\n".join(msg))
