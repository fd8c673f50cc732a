"""
Flat-faced bodies: triangle (2-D), triangular prism, tetrahedron and square pyramid (3-D).

API mirror of the reference's ``geometry/triangle_geometry.py``, ``prism_geometry.py``, ``tetrahedron_geometry.py`` and
``pyramid_geometry.py`` (same constructor arguments, checks, ``type`` tags, ``main_width`` / ``center`` definitions and
per-node predicates).  Constructor-time preparation (edge vectors, inward face normals, the split of a pyramid into two
tetrahedra) runs once on the host with the same torch operations the reference uses, so that the numbers handed to the
device kernels (``s3_mask_triangle`` / ``s3_mask_prism`` / ``s3_mask_tetrahedra``, include/s3hip.h) are the
reference's; ``check_cell`` is the single-cell host entry with the reference's truth table.
"""
import logging
from typing import List, Union

import torch as pt
from torch import Tensor

from .geometry_base import GeometryObject

logger = logging.getLogger(__name__)


class _Polytope(GeometryObject):
    """shared read-only attributes of the four bodies"""
    _type = ""

    type = property(lambda self: self._type)
    main_width = property(lambda self: self._main_width)
    center = property(lambda self: self._center)

    def check_cell(self, cell_nodes: Tensor, refine_geometry: bool = False) -> bool:
        return self._apply_mask(self._inside(cell_nodes), refine_geometry)


def _edge_sign(a: Tensor, b: Tensor) -> Tensor:
    """z-component of ``a x b`` for one 2-D vector ``a`` and a stack of 2-D vectors ``b``"""
    return a[0] * b[:, 1] - a[1] * b[:, 0]


class TriangleGeometry(_Polytope):
    """reference: triangle_geometry.py:15-199; predicate :80-103 (outline counts as inside)"""
    __short_description__ = "triangles (2D)"
    _type = "triangle"

    def __init__(self, name: str, keep_inside: bool, points: Union[list, Tensor], refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        for i, p in enumerate(points):
            if type(p) != Tensor:
                try:
                    points[i] = pt.tensor(p)
                except TypeError:
                    logger.error(f"Could not convert coordinate {i} of type {type(p)} to a tensor.")
            points[i] = points[i].type(pt.float64)       # the sign test needs floating point
        self._points = points
        self._check_geometry()
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()

    def _compute_main_width(self) -> float:
        corners = pt.stack(list(self._points))
        return (corners.max(0).values - corners.min(0).values).abs().max().item()

    def _compute_center(self) -> Tensor:
        return pt.stack(list(self._points)).mean(0)

    def _inside(self, vertices: Tensor) -> Tensor:
        p0, p1, p2 = self._points
        d1 = _edge_sign(p1 - p0, vertices - p0)
        d2 = _edge_sign(p2 - p1, vertices - p1)
        d3 = _edge_sign(p0 - p2, vertices - p0)
        mixed = ((d1 < 0) | (d2 < 0) | (d3 < 0)) & ((d1 > 0) | (d2 > 0) | (d3 > 0))
        return ~mixed

    def check_triangle(self, vertices: Tensor) -> Tensor:
        return self._inside(vertices)

    def kernel_spec(self) -> tuple:
        return "triangle", pt.stack(list(self._points)).numpy()

    def _check_geometry(self) -> None:
        assert isinstance(self._points, (list, Tensor)), (f"Expected the points to be a list or pt.Tensor, but found "
                                                          f"type {type(self._points)} instead.")
        assert len(self._points) == 3, f"Expected 3 points, but found {len(self._points)} points instead."
        assert all(len(p) == 2 for p in self._points), ("All given coordinates have to contain exactly 2 entries with "
                                                        "the x- and y-coordinates.")
        a, b = self._points[1] - self._points[0], self._points[2] - self._points[0]
        area = 0.5 * abs(a[0] * b[1] - a[1] * b[0])
        assert area > 0, f"The area of the triangle has to be larger than zero. Found an area of {area}."


class PrismGeometry3D(_Polytope):
    """two congruent triangles joined along a coordinate direction; reference: prism_geometry.py:12-201, predicate
    :90-118"""
    __short_description__ = "prisms (3D)"
    _type = "prism"

    def __init__(self, name: str, keep_inside: bool, positions: List[List[Union[list, tuple]]], refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._positions = positions
        self._check_geometry()
        self._positions = [pt.tensor(tri, dtype=pt.float64) for tri in self._positions]
        self._axis = (self._positions[1][0] - self._positions[0][0]).type(pt.float64)
        self._norm = self._axis.norm()
        # the triangles lie in the plane of the two directions the extrusion axis has no component in
        self._dim = pt.where(self._axis == 0)[0]
        not_aligned = "The specified triangles are not aligned along a coordinate direction."
        assert len(self._dim) == 2, not_aligned
        assert pt.allclose(self._positions[0][:, self._dim], self._positions[1][:, self._dim]), not_aligned
        self._triangles = [TriangleGeometry(f"{name}_{tag}", keep_inside=True, points=tri[:, self._dim])
                           for tag, tri in zip(("first", "second"), self._positions)]
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()

    def _compute_main_width(self) -> float:
        return max(self._axis.norm().item(), max(t.main_width for t in self._triangles))

    def _inside(self, vertices: Tensor) -> Tensor:
        offset = (vertices - self._positions[0][0]).type(self._axis.dtype)
        along = (offset * self._axis.expand_as(offset)).sum(-1) / self._norm
        in_height = pt.logical_and(0 <= along, along <= self._norm)
        return pt.logical_and(in_height, self._triangles[0].check_triangle(vertices[:, self._dim]))

    def kernel_spec(self) -> tuple:
        return ("prism", self._positions[0][0].numpy(), self._axis.numpy(), float(self._norm),
                [int(d) for d in self._dim], pt.stack(list(self._triangles[0]._points)).numpy())

    def _check_geometry(self) -> None:
        assert self._positions, "Found empty list for the positions. Please provide values for the prism."
        assert len(self._positions) == 2, (f"Expected exactly two triangles for the prism but found "
                                           f"{len(self._positions)} entries.")
        assert all(len(tri) == 3 for tri in self._positions), "Each triangle must have exactly 3 vertices."

    def _compute_center(self) -> Tensor:
        in_plane = pt.stack([t.center for t in self._triangles], -1).mean(1)
        along = self._axis.nonzero()[0]
        if len(along) > 1:
            raise NotImplementedError("The triangles are not aligned along a coordinate axis, which is currently not"
                                      " supported.")
        a = along.item()
        mid = pt.zeros((3,), dtype=self._axis.dtype)
        mid[a] = (self._positions[1][0, a] + self._positions[0][0, a]) / 2
        mid[self._dim] = in_plane
        return mid


class TetrahedronGeometry3D(_Polytope):
    """reference: tetrahedron_geometry.py:14-226; inward face normals :66-104, predicate :121-140"""
    __short_description__ = "tetrahedrons (3D)"
    _type = "tetrahedron"

    def __init__(self, name: str, keep_inside: bool, positions: Union[list, Tensor], refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._positions, self._normals = positions, None
        self._check_geometry()
        if isinstance(self._positions, Tensor):
            self._positions = self._positions.type(pt.float64)
        else:
            self._positions = pt.tensor(self._positions, dtype=pt.float64)
        homogeneous = pt.cat([self._positions, pt.ones((4, 1), dtype=pt.float64)], dim=1)
        assert abs(1 / 6 * pt.det(homogeneous)) > 0, "The tetrahedron provided has a volume of zero."
        self._compute_normals()
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()

    def _compute_main_width(self) -> float:
        return (self._positions.max(dim=0).values - self._positions.min(dim=0).values).max().item()

    def _compute_center(self) -> Tensor:
        return self._positions.mean(dim=0)

    def _compute_normals(self) -> None:
        """one normal per corner p (column p): the face it is tested against, turned towards the centroid"""
        a, b, c, d = self._positions
        faces = [pt.cross(b - a, c - a, dim=0), pt.cross(b - a, d - a, dim=0), pt.cross(c - a, d - a, dim=0),
                 pt.cross(c - b, d - c, dim=0)]
        normals = pt.stack(faces, dim=1)
        centroid = self._positions.mean(dim=0)
        towards = [pt.dot(centroid - self._positions[p, :], normals[:, p]) for p in range(4)]
        normals[:, pt.where(pt.tensor(towards) < 0)[0]] *= -1
        self._normals = normals

    def _inside(self, vertices: Tensor) -> Tensor:
        to_corner = vertices.unsqueeze(1) - self._positions.unsqueeze(0)
        seen = pt.tensor([[pt.dot(to_corner[v, p, :], self._normals[:, p]) for p in range(4)]
                          for v in range(vertices.size(0))])
        return ~(seen < 0).bool().any(1)

    def check_tetrahedron(self, vertices: Tensor) -> Tensor:
        return self._inside(vertices)

    def kernel_spec(self) -> tuple:
        return "tetrahedra", self._positions.numpy()[None], self._normals.numpy()[None]

    def _check_geometry(self) -> None:
        if isinstance(self._positions, list):
            assert self._positions, "Found empty list for the positions. Please provide values for the tetrahedron."
        else:
            assert isinstance(self._positions, Tensor), (f"Expected the points to be either a list, tuple or pt.Tensor,"
                                                         f" but found type {type(self._positions)}.")
        assert len(self._positions) == 4, (f"Expected exactly four points for the tetrahedron but found "
                                           f"{len(self._positions)} entries.")
        assert all(len(p) == 3 for p in self._positions), "Each point must have exactly 3 coordinates (x, y, z)."


class PyramidGeometry3D(_Polytope):
    """square pyramid given by five nodes in any order, handled as two tetrahedra sharing the main diagonal of the base
    and the apex; reference: pyramid_geometry.py:12-240 (apex :63-108, diagonal :110-131, predicate :156-170)"""
    __short_description__ = "square pyramids (3D)"
    _type = "pyramid"

    def __init__(self, name: str, keep_inside: bool, nodes: List[Union[list, tuple]], refine: bool = False,
                 min_refinement_level: int = None):
        super().__init__(name, keep_inside, refine, min_refinement_level)
        self._nodes = nodes
        self._check_geometry()
        self._nodes = pt.tensor(self._nodes)
        self._apex_idx = self._find_apex()
        self._diagonal_idx, self._off_diagonal = self._find_main_diagonal()
        halves = ([self._diagonal_idx[0], self._off_diagonal[0], self._diagonal_idx[1], self._apex_idx],
                  [self._diagonal_idx[1], self._off_diagonal[1], self._diagonal_idx[0], self._apex_idx])
        self._tets = [TetrahedronGeometry3D(f"tet{i}", self._keep_inside, self._nodes[ids])
                      for i, ids in enumerate(halves)]
        self._main_width = self._compute_main_width()
        self._center = self._compute_center()

    def _compute_main_width(self) -> float:
        return max(t.main_width for t in self._tets)

    def _compute_center(self) -> Tensor:
        return pt.stack([t.center for t in self._tets], -1).mean(1)

    def _find_apex(self) -> int:
        """the node farthest from the plane that holds most of the nodes (first such plane in (i, j, k) order)"""
        nodes, most, normal, on_plane = self._nodes, 0, None, None
        count = nodes.size(0)
        for i in range(count):
            for j in range(i + 1, count):
                for k in range(j + 1, count):
                    n = pt.cross(nodes[j] - nodes[i], nodes[k] - nodes[i], dim=0)
                    if n.norm() < 1e-12:        # collinear triple
                        continue
                    n /= n.norm()
                    hits = (abs((nodes - nodes[i]) @ n) < 1e-6).sum()
                    if hits > most:
                        most, normal, on_plane = hits, n, nodes[i]
        if normal is None:
            raise RuntimeError("No valid plane detected: the vertices may be collinear.")
        return pt.argmax(abs((nodes - on_plane) @ normal)).item()

    def _find_main_diagonal(self):
        base = [i for i in range(self._nodes.size(0)) if i != self._apex_idx]
        corners = self._nodes[base]
        dist2 = ((corners[:, None, :] - corners[None, :, :]) ** 2).sum(-1)
        dist2.fill_diagonal_(-float("inf"))
        i, j = pt.nonzero(dist2 == dist2.max(), as_tuple=True)
        diagonal = (base[i[0].item()], base[j[0].item()])
        return diagonal, [n for n in base if n not in diagonal]

    def _inside(self, vertices: Tensor) -> Tensor:
        return pt.stack([t.check_tetrahedron(vertices) for t in self._tets], dim=1).any(dim=1)

    def kernel_spec(self) -> tuple:
        return ("tetrahedra", pt.stack([t._positions for t in self._tets]).numpy(),
                pt.stack([t._normals for t in self._tets]).numpy())

    def _check_geometry(self) -> None:
        assert len(self._nodes) == 5, (f"The pyramid must have exactly five vertices but found {len(self._nodes)} "
                                       f"vertices.")
        for i, v in enumerate(self._nodes):
            assert isinstance(v, (list, tuple)), (f"Expected each vertex to be of type list or tuple but found "
                                                  f"type {type(v)} for vertex no. {i}.")
            assert len(v) == 3, (f"Expected each vertex to have exactly 3 components but found {len(v)} components "
                                 f"for entry {i}.")
