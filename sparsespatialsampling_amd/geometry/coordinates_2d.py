"""Arbitrary closed 2-D outlines given as coordinates.  API mirror of the reference's geometry/coordinates_2d.py.

The reference delegates to shapely (``Point.within(Polygon)``, coordinates_2d.py:70 = strictly inside, boundary
excluded).  This package carries its own polygon predicate so that the same arithmetic runs on the host
(``check_cell``) and in the ``s3_mask_polygon`` kernel; shapely is not required.
"""
from typing import Union

import numpy as np
from numpy import ndarray
from torch import Tensor, tensor

from .cube_geometry import mask_box
from .geometry_base import GeometryObject


class _Outline:
    """Closed polygon; ``strictly_inside`` is a crossing-number test in which boundary points count as outside."""

    def __init__(self, coordinates):
        xy = np.asarray(coordinates, dtype=np.float64)
        assert xy.ndim == 2 and xy.shape[1] == 2 and xy.shape[0] >= 3, "Expected at least three [x, y] coordinates."
        if np.all(xy[0] == xy[-1]):
            xy = xy[:-1]
        self.xy = np.ascontiguousarray(xy)
        self.bounds = (xy[:, 0].min(), xy[:, 1].min(), xy[:, 0].max(), xy[:, 1].max())

    @property
    def is_closed(self) -> bool:
        return self.xy.shape[0] >= 3

    def strictly_inside(self, px: float, py: float) -> bool:
        xi, yi = self.xy[:, 0], self.xy[:, 1]
        xj, yj = np.roll(xi, -1), np.roll(yi, -1)
        cross = (xj - xi) * (py - yi) - (yj - yi) * (px - xi)
        on_edge = ((cross == 0.0) & (np.minimum(xi, xj) <= px) & (px <= np.maximum(xi, xj)) &
                   (np.minimum(yi, yj) <= py) & (py <= np.maximum(yi, yj)))
        if on_edge.any():
            return False
        straddle = (yi > py) != (yj > py)
        with np.errstate(divide="ignore", invalid="ignore"):
            x_int = xi + (py - yi) * (xj - xi) / (yj - yi)
        return bool(np.count_nonzero(straddle & (px < x_int)) % 2)


class GeometryCoordinates2D(GeometryObject):
    __short_description__ = "2D coordinates for geometries"

    def __init__(self, name: str, keep_inside: bool, coordinates: Union[list, ndarray], refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._coordinates = _Outline(coordinates)
        self._type = "coord_2D"
        self._lower_bound = list(self._coordinates.bounds)[:2]
        self._upper_bound = list(self._coordinates.bounds)[2:]
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()
        self._check_geometry()

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        mask = tensor([self._coordinates.strictly_inside(float(cell_nodes[i, 0]), float(cell_nodes[i, 1]))
                       for i in range(cell_nodes.size(0))])
        return self._apply_mask(mask, refine_geometry)

    def pre_check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        return self._apply_mask(mask_box(cell_nodes, self._lower_bound, self._upper_bound), refine_geometry)

    def kernel_spec(self) -> tuple:
        return "polygon", self._coordinates.xy

    def _check_geometry(self) -> None:
        assert self._coordinates.is_closed, (f"Expected an enclosed area formed by the provided coordinates for "
                                             f"geometry {self.name}.")

    @property
    def type(self) -> str:
        return self._type

    @property
    def main_width(self) -> float:
        return self._main_width

    @property
    def center(self) -> Tensor:
        return self._center

    def _compute_main_width(self) -> float:
        return max([abs(u - l) for l, u in zip(self._lower_bound, self._upper_bound)])

    def _compute_center(self) -> Tensor:
        return (tensor(self._lower_bound) + tensor(self._upper_bound)) / 2.0
