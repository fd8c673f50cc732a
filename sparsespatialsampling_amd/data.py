"""
HDF5 / XDMF sink and loader of the S^3 results -- the classes and signatures of the reference's
``sparseSpatialSampling/data.py`` (``Dataloader`` :22-300, ``Datawriter`` :303-501, ``XDMFWriter`` :504-777) on the
package's own HDF5 layer (``h5io``: libs3h5.so on the HDF5 C library, h5py as a fallback).  On-disk layout and XDMF2
text are the reference's, so files written by either implementation load in the other and in ParaView:

    grid/{faces, vertices, centers}
    constant/{levels, metric, size_initial_cell, ...}
    data/<write time>/<field>_{center, vertices}

Organisation (this file's own): one ``_Placement`` rule table says where a ``write_data`` call lands; the XDMF writer
works from a shape inventory of the file (no array is loaded to describe it).  Where the reference ends the process
(``exit()``: unknown group, missing file, missing grid) this module raises ``ValueError`` / ``FileNotFoundError`` /
``KeyError`` instead -- a caller's process is not ours to end.
"""
import logging
from os.path import isfile, join
from typing import List, Union

import numpy as np
import torch as pt

from .const import CENTERS, CONST, DATA, FACES, GRID, VERTICES
from .h5io import open_h5

logger = logging.getLogger(__name__)


# ======================================================================================================================
# loader
# ======================================================================================================================
class Dataloader:
    """lazy reader of an S^3 HDF5 file.  Names follow the reference: ``vertices`` are the cell CENTRES, ``nodes`` the cell
    corners, ``weights`` the cell areas / volumes."""

    # attribute -> (dataset, squeeze)
    _ARRAYS = {"vertices": (f"{GRID}/{CENTERS}", False), "nodes": (f"{GRID}/{VERTICES}", False),
               "faces": (f"{GRID}/{FACES}", False), "levels": (f"{CONST}/levels", True), "metric": (f"{CONST}/metric", True)}

    def __init__(self, load_path: str, file_name: str, dtype: pt.dtype = pt.float32):
        self._load_path, self._file_name, self._dtype = load_path, file_name, dtype
        self._cache = {}
        self._scan(tolerant=True)

    def _open(self):
        return open_h5(join(self._load_path, self._file_name), "r")

    def _scan(self, tolerant: bool) -> None:
        """grid size and initial cell width (a file without the width only fails later, as in the reference)"""
        self._cache = {}
        with self._open() as f:
            self._n_cells, self._n_dimensions = f.shape(f"{GRID}/{CENTERS}")
            if f.exists(f"{CONST}/size_initial_cell"):
                self._size_initial_cell = f.read(f"{CONST}/size_initial_cell")[()]
            elif tolerant:
                logger.warning("Could not load initial cell size.")
            else:
                raise KeyError(f"{CONST}/size_initial_cell")

    def _array(self, name: str) -> pt.Tensor:
        if name not in self._cache:
            key, squeeze = self._ARRAYS[name]
            with self._open() as f:
                t = pt.from_numpy(f.read(key))
            self._cache[name] = t.squeeze() if squeeze else t
        return self._cache[name]

    vertices = property(lambda self: self._array("vertices"))
    nodes = property(lambda self: self._array("nodes"))
    faces = property(lambda self: self._array("faces"))
    levels = property(lambda self: self._array("levels"))
    metric = property(lambda self: self._array("metric"))

    @property
    def write_times(self) -> List[str]:
        if "write_times" not in self._cache:
            with self._open() as f:
                self._cache["write_times"] = f.keys(DATA) if DATA in f.keys() else None
        return self._cache["write_times"]

    @property
    def field_names(self) -> dict:
        """{write time: [fields with a ``_center`` dataset]}"""
        if "field_names" not in self._cache:
            with self._open() as f:
                self._cache["field_names"] = {t: [n.split("_")[0] for n in f.keys(f"{DATA}/{t}") if n.endswith("center")]
                                              for t in f.keys(DATA)}
        return self._cache["field_names"]

    @property
    def weights(self) -> pt.Tensor:
        """cell areas (2-D) / volumes (3-D): (size_initial_cell / 2^level)^d  (reference data.py:240-247)"""
        if "weights" not in self._cache:
            self._cache["weights"] = (pow(self._size_initial_cell / pow(2, self.levels), self._n_dimensions)).squeeze()
        return self._cache["weights"]

    @property
    def load_path(self) -> str:
        return self._load_path

    @load_path.setter
    def load_path(self, value: str) -> None:
        self._load_path = value
        self._scan(tolerant=False)

    @property
    def file_name(self) -> str:
        return self._file_name

    @file_name.setter
    def file_name(self, value: str) -> None:
        self._file_name = value
        self._scan(tolerant=False)

    def load_snapshot(self, field_name: Union[List[str], str],
                      write_times: Union[str, List[str]] = None) -> Union[List[pt.Tensor], pt.Tensor]:
        """data matrix ``[N_cells, (N_dims,) N_times]`` of one field (cell-centred values), or a list of them"""
        times = self.write_times if write_times is None else write_times
        times = [times] if isinstance(times, str) else list(times)
        fields = [field_name] if isinstance(field_name, str) else list(field_name)
        out = []
        with self._open() as f:
            for name in fields:
                per_time = [pt.from_numpy(f.read(f"{DATA}/{t}/{name}_center")) for t in times]
                out.append(pt.stack(per_time, dim=-1).to(self._dtype))
        return out[0] if len(out) == 1 else out


# ======================================================================================================================
# writer
# ======================================================================================================================
class _Placement:
    """where a ``write_data`` call lands: HDF5 prefix and whether the dataset name gets a location suffix"""

    def __init__(self, prefix_of, suffixed, duplicate_message):
        self.prefix_of, self.suffixed, self.duplicate_message = prefix_of, suffixed, duplicate_message


_PLACEMENTS = {
    GRID: _Placement(lambda t: GRID, False, None),                       # the grid is written once; a duplicate is an error
    CONST: _Placement(lambda t: CONST, False, "Field {name} already exists in time step {t}. Skipping field {name}."),
    DATA: _Placement(lambda t: f"{DATA}/{t}", True, "Field {name} already exists in the HDF file. Skipping field {name}."),
}


class Datawriter:
    def __init__(self, file_path: str, file_name: str, mode: str = "w", mixed: bool = False):
        self._file_path, self._file_name, self._mode, self._mixed = file_path, file_name, mode, mixed
        self._file = open_h5(join(file_path, file_name), mode)
        self._n_cells = None

    def close(self) -> None:
        if self._file is not None:
            skipped = self._file.flush()
            if skipped:
                logger.warning(f"{skipped} dataset(s) of the last batches existed already in {self._file_name} and were skipped.")
            self._file.close()
            self._file = None

    def write_grid(self, loader: Dataloader) -> None:
        """copy the grid of an existing S^3 file (centres, corner nodes, faces)"""
        self._n_cells = loader.vertices.shape[0]
        for key, values in ((CENTERS, loader.vertices), (VERTICES, loader.nodes), (FACES, loader.faces)):
            self.write_data(key, group=GRID, data=values)

    def _resolve(self, name: str, data, group: str, time_step):
        """(HDF5 path, placement) of one ``write_data`` call -- reference data.py:361-430"""
        if time_step is not None and group != DATA:
            group = DATA                                      # a time step always means temporal data
        if group not in _PLACEMENTS:
            raise ValueError(f"Unknown group type {group!r}, available types are '{DATA}', '{CONST}' and '{GRID}'.")
        place = _PLACEMENTS[group]
        if group == DATA and time_step is None:
            logger.warning(f"No time step for group 'data' provided. Writing data to the zeroth time step '{DATA}/0'.")
            time_step = "0"
        if place.suffixed and self._n_cells is not None and not name.endswith(("center", "vertices")):
            # with a known grid (write_grid was used) plain names get the location suffix the loader expects
            name = f"{name}_center" if np.shape(data)[0] == self._n_cells else f"{name}_vertices"
        return f"{place.prefix_of(time_step)}/{name}", place, name, time_step

    def write_data(self, name: str, data: any, group: str = "constant", time_step: Union[int, float, str] = None) -> None:
        """one dataset in ``grid/``, ``constant/`` or ``data/<time_step>/``"""
        path, place, name, time_step = self._resolve(name, data, group, time_step)
        if not self._file.write(path, data):
            if place.duplicate_message is None:
                raise ValueError(f"Unable to create dataset {path!r}: it exists already.")
            logger.warning(place.duplicate_message.format(name=name, t=time_step))

    def write_snapshots(self, name: str, times: list, host_snapshot_major, ready=None) -> None:
        """``data/<times[i]>/<name>`` = ``host_snapshot_major[i]`` for a whole batch, queued and written in the background
        (this build's addition: ExportData hands over its snapshot-major download buffer).
        ``ready = (int32 host tensor, value)``: the buffer is still being filled by a device-to-host copy, complete once the
        tensor's first element reaches ``value``."""
        self._file.write_snapshots([str(t) for t in times], name, host_snapshot_major, group=DATA, ready=ready)

    def wait_buffer(self, host) -> None:
        """returns once no queued write reads from ``host`` any more (a closed file has nothing queued)"""
        if self._file is not None:
            self._file.wait_buffer(host)

    def write_xdmf_file(self) -> None:
        if self._file is not None:
            self.close()
        full = join(self._file_path, self._file_name)
        if not isfile(full):
            raise FileNotFoundError(f"Could not find {full}. Make sure the file exists and the provided path is correct.")
        logger.info(f"Writing XDMF file for file {self._file_name}")
        XDMFWriter(self._file_path, self._file_name, mixed=self._mixed).write_xdmf()

    @property
    def mode(self) -> str:
        return self._mode

    @mode.setter
    def mode(self, value) -> None:
        """re-open the file in another mode (the export path switches a finished file to "a" for the next field)"""
        if self._file is not None:
            self.close()
        self._mode = value
        self._file = open_h5(join(self._file_path, self._file_name), value)

    @property
    def file_name(self) -> str:
        return self._file_name

    @property
    def is_open(self) -> bool:
        return self._file is not None

    @property
    def n_cells(self) -> Union[int, None]:
        return self._n_cells

    @n_cells.setter
    def n_cells(self, value: int) -> None:
        self._n_cells = value


# ======================================================================================================================
# XDMF
# ======================================================================================================================
class XDMFWriter:
    """XDMF2 description of an S^3 HDF5 file: a temporal collection, or one uniform grid when there is no ``data`` group.
    The text is the reference writer's (data.py:566-777); it is assembled here from a shape inventory of the file."""

    _OPEN = '<?xml version="1.0"?>\n<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>\n<Xdmf Version="2.0">\n'

    def __init__(self, file_path: str, file_name: str, grid_name: str = "grid_s_cube", mixed: bool = False):
        self._file_path, self._hdf_file_name, self._grid_name, self._mixed = file_path, file_name, grid_name, mixed
        self._xdmf_file_name = f"{file_name.split('.h5')[0]}.xdmf"
        self._inventory = self._take_inventory()
        grid = self._inventory[GRID]
        self._n_cells, self._n_dimensions = grid[CENTERS][0], grid[CENTERS][-1]
        self._n_faces, self._n_vertices = grid[FACES][0], grid[VERTICES][0]
        self._grid_type = "Mixed" if mixed else ("Quadrilateral" if self._n_dimensions == 2 else "Hexahedron")
        self._dims = "XY" if self._n_dimensions == 2 else "XYZ"

    def _take_inventory(self) -> dict:
        """{group: {dataset: shape}} with ``data`` as {time: {dataset: shape}}; raises when the grid is incomplete"""
        inv = {}
        with open_h5(join(self._file_path, self._hdf_file_name), "r") as f:
            top = f.keys()
            if GRID not in top:
                raise KeyError("Found no grid in the provided HDF5 file. Unable to create XDMF file without a grid.")
            have = f.keys(GRID)
            for key, what in ((FACES, "faces"), (CENTERS, "centers"), (VERTICES, "vertices")):
                if key not in have:
                    raise KeyError(f"Unable to find cell {what} in group {GRID}. Make sure the key to the cell {what} is present "
                                   f"and named {key}.")
            inv[GRID] = {k: f.shape(f"{GRID}/{k}") for k in have}
            inv[CONST] = {k: f.shape(f"{CONST}/{k}") for k in f.keys(CONST)} if CONST in top else None
            inv[DATA] = ({t: {k: f.shape(f"{DATA}/{t}/{k}") for k in f.keys(f"{DATA}/{t}")} for t in f.keys(DATA)}
                         if DATA in top else None)
        return inv

    # -- text fragments --------------------------------------------------------------------------------------------
    def _mesh(self) -> str:
        face_dims = f"{self._n_faces}" if self._mixed else f"{self._n_faces} {pow(2, self._n_dimensions)}"
        h5 = self._hdf_file_name
        return (f'<Topology TopologyType="{self._grid_type}" NumberOfElements="{self._n_faces}">\n'
                f'<DataItem Format="HDF" DataType="Int" Dimensions="{face_dims}">\n{h5}:/{GRID}/{FACES}\n</DataItem>\n'
                f'</Topology>\n<Geometry GeometryType="{self._dims}">\n'
                f'<DataItem Rank="2" Dimensions="{self._n_vertices} {self._n_dimensions}" NumberType="Float" Precision="8" '
                f'Format="HDF">\n{h5}:/{GRID}/{VERTICES}\n</DataItem>\n</Geometry>\n')

    def _field(self, label: str, h5_key: str, shape: tuple) -> str:
        """cell- or node-centred attribute; '' (with a warning) when the leading size matches neither"""
        location = {self._n_cells: "Cell", self._n_vertices: "Node"}.get(shape[0] if shape else None)
        if location is None:
            logger.warning(f"Field in '{h5_key}' with a size of {shape} doesn't match the number of cells with N_cells = "
                           f"{self._n_cells} or the number of vertices with N_vertices = {self._n_vertices}. Skipping this field.")
            return ""
        width = 1 if len(shape) == 1 else shape[1]
        return (f'<Attribute Name="{label}" AttributeType="Vector" Center="{location}">\n<DataItem NumberType="Float" '
                f'Precision="8" Format="HDF" Dimensions="{shape[0]} {width}">\n{self._hdf_file_name}:/{h5_key}\n</DataItem>\n'
                f'</Attribute>\n')

    def _constant_fields(self) -> str:
        const = self._inventory[CONST]
        if const is None:
            logger.info("Couldn't find any constant fields to write.")
            return ""
        return "".join(self._field(k, f"{CONST}/{k}", shape) for k, shape in const.items()
                       if shape and shape[0] in (self._n_cells, self._n_vertices))

    # -- documents -------------------------------------------------------------------------------------------------
    def write_xdmf(self) -> None:
        body = self._temporal_collection() if self._inventory[DATA] is not None else self._single_grid()
        with open(join(self._file_path, self._xdmf_file_name), "w") as out:
            out.write(self._OPEN + body)

    def _single_grid(self) -> str:
        return (f'<Domain>\n<Grid Name="{self._grid_name}" GridType="Uniform">\n' + self._mesh() + self._constant_fields()
                + "</Grid>\n</Domain>\n</Xdmf>")

    def _temporal_collection(self) -> str:
        parts = [f'<Domain>\n<Grid Name="{self._grid_name}" GridType="Collection" CollectionType="temporal">\n']
        data = self._inventory[DATA]
        for i, t in enumerate(sorted(data, key=float)):
            parts.append(f'<Grid Name="{self._grid_name} {t}" GridType="Uniform">\n<Time Value="{t}"/>\n' + self._mesh())
            if i == 0:                                        # constant fields go with the first time step
                parts.append(self._constant_fields())
            for k, shape in data[t].items():
                stem = k.split("_")                           # <field>_<center|vertices>
                parts.append(self._field("_".join(stem[:-1]) if len(stem) > 1 else k, f"{DATA}/{t}/{k}", shape))
            parts.append("</Grid>\n")
        parts.append("</Grid>\n</Domain>\n</Xdmf>")
        return "".join(parts)
