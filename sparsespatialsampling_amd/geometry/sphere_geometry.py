"""Circles (2-D) / spheres (3-D).  API mirror of the reference's geometry/sphere_geometry.py."""
from typing import Union

from torch import Tensor, tensor, as_tensor

from .geometry_base import GeometryObject


def mask_sphere(vertices: Tensor, center: list, radius: float) -> Tensor:
    """``||x - center||_2 <= radius`` (inclusive; pinned by the reference's tests/test_sphere_geometry.py:43-75).
    Stands in for flowtorch.data.mask_sphere (sphere_geometry.py:69)."""
    return (vertices - as_tensor(center, dtype=vertices.dtype)).norm(dim=1) <= radius


class SphereGeometry(GeometryObject):
    __short_description__ = "circles (2D) or spheres (3D)"

    def __init__(self, name: str, keep_inside: bool, position: list, radius: Union[int, float], refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._position = position
        self._radius = radius
        self._type = "sphere"
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()
        self._check_geometry()

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        assert cell_nodes.size(1) == len(self._position), (
            f"Number of dimensions of the cell does not match the number of dimensions for the position. Expected "
            f"{cell_nodes.size(-1)} values, found {len(self._position)} for geometry {self.name}.")
        return self._apply_mask(mask_sphere(cell_nodes, self._position, self._radius), refine_geometry)

    def kernel_spec(self) -> tuple:
        return "sphere", [float(v) for v in self._position], float(self._radius)

    def _check_geometry(self) -> None:
        assert self._position, "Found empty list for the position. Please provide values for the position."
        assert isinstance(self._radius, (int, float)), (f"Expected the type of radius to be Union[int, float], got "
                                                        f"{type(self._radius)} for geometry {self.name} instead.")
        assert self._radius > 0, f"Expected a radius larger than zero but found a value of {self._radius}."

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
        return float(self._radius)

    def _compute_center(self) -> Tensor:
        return tensor(self._position)
