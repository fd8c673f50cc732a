"""Axis-aligned rectangles (2-D) / boxes (3-D).  API mirror of the reference's geometry/cube_geometry.py."""
from torch import Tensor, tensor, ones, bool as pt_bool

from .geometry_base import GeometryObject


def mask_box(vertices: Tensor, lower: list, upper: list) -> Tensor:
    """``lower[i] <= x_i <= upper[i]`` in every dimension (inclusive faces; pinned by the reference's
    tests/test_cube_geometry.py:46-78).  Stands in for flowtorch.data.mask_box (cube_geometry.py:71)."""
    mask = ones(vertices.shape[0], dtype=pt_bool)
    for i in range(len(lower)):
        mask &= (vertices[:, i] >= lower[i]) & (vertices[:, i] <= upper[i])
    return mask


class CubeGeometry(GeometryObject):
    __short_description__ = "rectangles (2D) or cubes (3D)"

    def __init__(self, name: str, keep_inside: bool, lower_bound: list, upper_bound: list, refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._lower_bound = lower_bound
        self._upper_bound = upper_bound
        self._type = "cube"
        self._check_geometry()
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        assert cell_nodes.size(-1) == len(self._lower_bound), (
            f"Number of dimensions of the cell does not match the number of given bounds. Expected "
            f"{cell_nodes.size(-1)} values, found {len(self._lower_bound)} for geometry {self.name}.")
        return self._apply_mask(mask_box(cell_nodes, self._lower_bound, self._upper_bound), refine_geometry)

    def kernel_spec(self) -> tuple:
        return "box", [float(v) for v in self._lower_bound], [float(v) for v in self._upper_bound]

    def _check_geometry(self) -> None:
        assert self._lower_bound, "Found empty list for the lower bound. Please provide values for the lower bound."
        assert self._upper_bound, "Found empty list for the upper bound. Please provide values for the upper bound."
        assert len(self._lower_bound) == len(self._upper_bound), (
            f"The number of provided boundaries for the lower bound does not match the number of boundaries for the "
            f"upper bound. Found {len(self._lower_bound)} values for the lower bound but {len(self._upper_bound)} "
            f"values for the upper bound for geometry {self.name}.")
        for i, (lo, hi) in enumerate(zip(self._lower_bound, self._upper_bound)):
            assert lo < hi, (f"Value of {lo} for the lower bound at position {i} is larger or equal than the value of "
                             f"{hi} for the upper bound for geometry {self.name}. The the lower bound must be smaller "
                             f"than the upper bound!")

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
