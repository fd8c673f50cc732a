"""Cylinders, truncated cones and cones in 3-D, given by the two end points of the axis and one radius (cylinder) or one
radius per end point (cone).

Behaves like the reference's ``geometry/cylinder_geometry.py`` (signatures, attribute names, ``AssertionError`` on bad
arguments: reference lines 85-124; the end points pass through float32 before the axis is taken in float64: lines 51-56;
predicate: lines 126-157).  The per-cell predicate of the refine loop is the device kernel ``s3_mask_cylinder`` fed from
``kernel_spec()`` with exactly the numbers kept here."""
from numbers import Real
from typing import List, Union

from torch import Tensor, as_tensor, cross, float32, float64

from .geometry_base import GeometryObject


class CylinderGeometry3D(GeometryObject):
    __short_description__ = "cylinders, conical objects and cones (3D)"
    _type = "cylinder"

    def __init__(self, name: str, keep_inside: bool, position: List[Union[list, tuple]],
                 radius: Union[int, float, list, tuple], refine: bool = False, min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._position, self._radius = position, radius
        self._validate()
        # end points in float32 (what `tensor(...).float()` of the reference leaves), axis and its length in float64
        self._position = as_tensor(self._position, dtype=float32)
        self._axis = (self._position[1] - self._position[0]).type(float64)
        self._norm = self._axis.norm()
        self._main_width, self._center = self._compute_main_width(), self._compute_center()

    def _is_cone(self) -> bool:
        return not isinstance(self._radius, Real)

    def _argument_rules(self) -> tuple:
        p, r = self._position, self._radius
        return (
            (lambda: len(p) == 2, lambda: f"the axis needs its two end points, got {len(p)} position(s)."),
            (lambda: list(p[0]) != list(p[1]), lambda: "both end points of the axis coincide (length zero)."),
            (lambda: isinstance(r, (Real, list, tuple)),
             lambda: f"radius has to be a number or a pair of numbers, got {type(r)}."),
            (lambda: r > 0 if isinstance(r, Real) else len(r) == 2,
             lambda: f"radius {r} has to be positive." if isinstance(r, Real) else f"a cone needs two radii, got {len(r)}."),
            (lambda: isinstance(r, Real) or (min(r) >= 0 and max(r) > 0),
             lambda: f"radii {r}: none may be negative and one has to be positive."),
            # (equal radii describe a cylinder; like the reference, a cone wants them to differ)
            (lambda: isinstance(r, Real) or r[0] != r[1], lambda: f"radii {r} are equal: pass a single radius for a cylinder."),
        )

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        return self._apply_mask(self._mask_cylinder(cell_nodes), refine_geometry)

    def kernel_spec(self) -> tuple:
        r0, r1 = (self._radius[0], self._radius[1]) if self._is_cone() else (self._radius, self._radius)
        return ("cylinder", self._position[0].double().tolist(), self._axis.tolist(), float(self._norm), float(r0), float(r1),
                int(self._is_cone()))

    def _mask_cylinder(self, vertices: Tensor) -> Tensor:
        """per vertex: between the two end planes and no farther from the axis than the radius there (linear between the
        two radii for a cone).  Same operations in the same order as cylinder_geometry.py:126-157 -- the device kernel and
        the goldens are pinned to these roundings."""
        from_p0 = (vertices - self._position[0]).type(float64)
        axis = self._axis.expand_as(from_p0)
        radial = cross(axis, from_p0, 1).norm(dim=1) / self._norm
        along = (from_p0 * axis).sum(-1) / self._norm
        allowed = self._radius[0] + along / self._norm * (self._radius[1] - self._radius[0]) if self._is_cone() else self._radius
        return (0 <= along) & (along <= self._norm) & (radial <= allowed)

    type = property(lambda self: self._type)
    main_width = property(lambda self: self._main_width)
    center = property(lambda self: self._center)

    def _compute_main_width(self) -> float:
        """the larger of the axis length and the (largest) radius"""
        widest = max(self._radius) if self._is_cone() else self._radius
        return max(widest, self._norm.item())

    def _compute_center(self) -> Tensor:
        return self._position.sum(0) / 2             # (midpoint of the two end points; float32 like them)
