"""Geometry objects that have a device predicate (box, sphere, cylinder / cone, closed 2-D outline) and their base."""
from . import coordinates_2d, cube_geometry, cylinder_geometry, geometry_base, sphere_geometry

GeometryObject = geometry_base.GeometryObject
CubeGeometry = cube_geometry.CubeGeometry
SphereGeometry = sphere_geometry.SphereGeometry
CylinderGeometry3D = cylinder_geometry.CylinderGeometry3D
GeometryCoordinates2D = coordinates_2d.GeometryCoordinates2D

__all__ = ["GeometryObject", "CubeGeometry", "SphereGeometry", "CylinderGeometry3D", "GeometryCoordinates2D"]
