"""HDF5 group / dataset names shared by the data loader and writer (same on-disk names as the reference's const.py)."""

CONST = "constant"     # time-independent cell fields
GRID = "grid"          # faces / centers / vertices
DATA = "data"          # one sub-group per write time

FACES = "faces"
CENTERS = "centers"
VERTICES = "vertices"
