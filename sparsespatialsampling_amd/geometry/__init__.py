"""Geometry objects that have a device predicate (box, sphere, cylinder / cone, closed 2-D outline, triangle, prism,
tetrahedron, square pyramid) and their base."""
from . import coordinates_2d, cube_geometry, cylinder_geometry, geometry_base, polytope_geometry, sphere_geometry

GeometryObject = geometry_base.GeometryObject
CubeGeometry = cube_geometry.CubeGeometry
SphereGeometry = sphere_geometry.SphereGeometry
CylinderGeometry3D = cylinder_geometry.CylinderGeometry3D
GeometryCoordinates2D = coordinates_2d.GeometryCoordinates2D
TriangleGeometry = polytope_geometry.TriangleGeometry
PrismGeometry3D = polytope_geometry.PrismGeometry3D
TetrahedronGeometry3D = polytope_geometry.TetrahedronGeometry3D
PyramidGeometry3D = polytope_geometry.PyramidGeometry3D

__all__ = ["GeometryObject", "CubeGeometry", "SphereGeometry", "CylinderGeometry3D", "GeometryCoordinates2D",
           "TriangleGeometry", "PrismGeometry3D", "TetrahedronGeometry3D", "PyramidGeometry3D"]
