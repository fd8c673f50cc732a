"""Axis-aligned rectangles (2-D) / boxes (3-D): the numerical domain of most cases, or a box-shaped body.

Behaves like the reference's ``geometry/cube_geometry.py`` (constructor and ``check_cell`` signatures, attribute names,
``AssertionError`` on bad arguments: reference lines 76-99; inclusive faces: its tests/test_cube_geometry.py:46-78).  The
per-cell predicate of the refine loop is the device kernel ``s3_mask_box`` fed from ``kernel_spec()``."""
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
    __short_description__ = "axis-aligned rectangles (2D) / boxes (3D)"
    _type = "cube"

    def __init__(self, name: str, keep_inside: bool, lower_bound: list, upper_bound: list, refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._lower_bound, self._upper_bound = lower_bound, upper_bound
        self._validate()
        self._main_width, self._center = self._compute_main_width(), self._compute_center()

    def _argument_rules(self) -> tuple:
        lo, hi = self._lower_bound, self._upper_bound
        inverted = [i for i in range(min(len(lo), len(hi))) if not lo[i] < hi[i]]
        return (
            (lambda: len(lo) > 0, lambda: "lower_bound holds no values."),
            (lambda: len(hi) > 0, lambda: "upper_bound holds no values."),
            (lambda: len(lo) == len(hi),
             lambda: f"lower_bound has {len(lo)} entries, upper_bound {len(hi)}; one per dimension is needed in both."),
            (lambda: not inverted,
             lambda: f"the box is empty along axis {inverted[0]} (lower {lo[inverted[0]]} >= upper {hi[inverted[0]]})."),
        )

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        n_dims = len(self._lower_bound)
        assert cell_nodes.size(-1) == n_dims, (f"Geometry '{self.name}' is {n_dims}-dimensional, the cell nodes have "
                                               f"{cell_nodes.size(-1)} coordinates.")
        return self._apply_mask(mask_box(cell_nodes, self._lower_bound, self._upper_bound), refine_geometry)

    def kernel_spec(self) -> tuple:
        return "box", [float(v) for v in self._lower_bound], [float(v) for v in self._upper_bound]

    type = property(lambda self: self._type)
    main_width = property(lambda self: self._main_width)
    center = property(lambda self: self._center)

    def _compute_main_width(self) -> float:
        """the longest edge (the width of the root cell when this box is the domain)"""
        return max(abs(hi - lo) for lo, hi in zip(self._lower_bound, self._upper_bound))

    def _compute_center(self) -> Tensor:
        return tensor([self._lower_bound, self._upper_bound]).sum(0) / 2.0
