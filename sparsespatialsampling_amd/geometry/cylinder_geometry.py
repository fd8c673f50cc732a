"""Cylinders, truncated cones and cones (3-D).  API mirror of the reference's geometry/cylinder_geometry.py."""
from typing import List, Union

from torch import Tensor, cross, float64, tensor

from .geometry_base import GeometryObject


class CylinderGeometry3D(GeometryObject):
    __short_description__ = "cylinders, conical objects and cones (3D)"

    def __init__(self, name: str, keep_inside: bool, position: List[Union[list, tuple]],
                 radius: Union[int, float, list, tuple], refine: bool = False, min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._position = position
        self._radius = radius
        self._type = "cylinder"
        self._check_geometry()
        # like the reference (cylinder_geometry.py:51-56) the end points are rounded through float32 before the axis
        # is widened to float64; the device kernel receives exactly these values
        self._position = tensor(self._position).float()
        self._axis = (self._position[1, :] - self._position[0, :]).type(float64)
        self._norm = self._axis.norm()
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        return self._apply_mask(self._mask_cylinder(cell_nodes), refine_geometry)

    def kernel_spec(self) -> tuple:
        if isinstance(self._radius, (int, float)):
            r0 = r1 = float(self._radius)
            cone = 0
        else:
            r0, r1, cone = float(self._radius[0]), float(self._radius[1]), 1
        return ("cylinder", self._position[0].double().tolist(), self._axis.tolist(), float(self._norm), r0, r1, cone)

    def _check_geometry(self) -> None:
        assert self._position, "Found empty list for the position. Please provide values for the positions."
        assert len(self._position) == 2, (f"Expected exactly two entries for the position but found "
                                          f"{len(self._position)} entries.")
        assert self._position[0] != self._position[1], ("Expected two different positions, a cylinder of length zero "
                                                        "is invalid.")
        assert isinstance(self._radius, (int, float, list, tuple)), (
            f"Expected the type of radius to be Union[int, float, list, tuple], got {type(self._radius)} for geometry "
            f"{self.name} instead.")
        if isinstance(self._radius, (int, float)):
            assert self._radius > 0, f"Expected a radius larger than zero but found a value of {self._radius}."
        else:
            assert len(self._radius) == 2, f"Expected two values for the radii but found {len(self._radius)}."
            assert self._radius[0] >= 0 and self._radius[1] >= 0, (f"Expected all radii >= 0 but found a values of "
                                                                   f"{self._radius}.")
            assert (self._radius[0] == self._radius[1]) == 0, (f"Both values for the radii can't be zero. At least "
                                                               f"one radius has to be > 0 but found values of "
                                                               f"{self._radius}.")

    def _mask_cylinder(self, vertices: Tensor) -> Tensor:
        """inside <=> 0 <= projection <= |axis| and normal distance <= local radius (cylinder_geometry.py:126-157)"""
        rel = (vertices - self._position[0, :]).type(self._axis.dtype)
        axis = self._axis.expand_as(rel)
        normal_distance = cross(axis, rel, 1).norm(dim=1) / self._norm
        projection = (rel * axis).sum(-1) / self._norm
        if isinstance(self._radius, (float, int)):
            local_radius = self._radius
        else:
            local_radius = self._radius[0] + projection / self._norm * (self._radius[1] - self._radius[0])
        return (0 <= projection) & (projection <= self._norm) & (normal_distance <= local_radius)

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
        return max(max(self._radius) if isinstance(self._radius, list) else self._radius, self._axis.norm().item())

    def _compute_center(self) -> Tensor:
        return self._position.mean(0)
