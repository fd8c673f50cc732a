from .coordinates_2d import GeometryCoordinates2D
from .cube_geometry import CubeGeometry
from .cylinder_geometry import CylinderGeometry3D
from .geometry_base import GeometryObject
from .sphere_geometry import SphereGeometry
