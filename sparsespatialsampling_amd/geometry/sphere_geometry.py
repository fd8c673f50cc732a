"""Circles (2-D) / spheres (3-D).  API mirror of the reference's geometry/sphere_geometry.py."""
from typing import Union

from torch import Tensor, as_tensor, tensor

from .geometry_base import GeometryObject


def mask_sphere(vertices: Tensor, center: list, radius: float) -> Tensor:
    """per-vertex ``||x - center||_2 <= radius`` (the surface counts as inside, pinned by the reference's
    tests/test_sphere_geometry.py:43-75); stands in for flowtorch.data.mask_sphere used at sphere_geometry.py:69"""
    offset = vertices - as_tensor(center, dtype=vertices.dtype)
    return offset.norm(dim=1) <= radius


class SphereGeometry(GeometryObject):
    __short_description__ = "circles (2D) or spheres (3D)"
    _type = "sphere"

    def __init__(self, name: str, keep_inside: bool, position: list, radius: Union[int, float], refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._position, self._radius = position, radius
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()
        self._check_geometry()

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        assert cell_nodes.size(1) == len(self._position), (
            f"Geometry {self.name} is {len(self._position)}-dimensional, the cell has {cell_nodes.size(-1)} dimensions.")
        return self._apply_mask(mask_sphere(cell_nodes, self._position, self._radius), refine_geometry)

    def kernel_spec(self) -> tuple:
        return "sphere", [float(v) for v in self._position], float(self._radius)

    def _check_geometry(self) -> None:
        assert self._position, "The position of the sphere is empty."
        assert isinstance(self._radius, (int, float)), (f"Geometry {self.name}: the radius has to be an int or float, "
                                                        f"got {type(self._radius)}.")
        assert self._radius > 0, f"The radius has to be positive, got {self._radius}."

    type = property(lambda self: self._type)
    main_width = property(lambda self: self._main_width)
    center = property(lambda self: self._center)

    def _compute_main_width(self) -> float:
        return float(self._radius)

    def _compute_center(self) -> Tensor:
        return tensor(self._position)
