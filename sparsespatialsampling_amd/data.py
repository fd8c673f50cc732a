"""
HDF5 / XDMF sink and loader of the S^3 results -- API mirror of the reference's ``sparseSpatialSampling/data.py``
(``Dataloader`` :22-300, ``Datawriter`` :303-501, ``XDMFWriter`` :504-777) with the same on-disk layout:

    grid/{faces, vertices, centers}
    constant/{levels, metric, size_initial_cell, ...}
    data/<write time>/<field>_{center, vertices}

and the same XDMF2 text (temporal collection, or a single uniform grid when there is no ``data`` group), so files written
by either implementation load in the other and in ParaView.  This is disk I/O: it stays on the
host and needs ``h5py``, which is imported lazily so that the rest of the package works without it.
"""
import logging
from os.path import isfile, join
from typing import List, Union

import torch as pt

from .const import CENTERS, CONST, DATA, FACES, GRID, VERTICES

logger = logging.getLogger(__name__)


def _h5file(*args, **kwargs):
    try:
        from h5py import File
    except ModuleNotFoundError as e:        # pragma: no cover - depends on the environment
        raise ModuleNotFoundError("h5py is required for reading / writing S^3 HDF5 files (pip install h5py)") from e
    return File(*args, **kwargs)


def _np(data):
    return data.detach().cpu().numpy() if isinstance(data, pt.Tensor) else data


class Dataloader:
    def __init__(self, load_path: str, file_name: str, dtype: pt.dtype = pt.float32):
        self._load_path = load_path
        self._file_name = file_name
        self._dtype = dtype
        self._read_header(strict=False)
        self._clear_cache()

    # -- helpers -------------------------------------------------------------------------------------------------
    def _path(self) -> str:
        return join(self._load_path, self._file_name)

    def _read(self, key: str):
        with _h5file(self._path(), "r") as f:
            return f.get(key)[()]

    def _read_header(self, strict: bool) -> None:
        with _h5file(self._path(), "r") as f:
            shape = f.get(f"{GRID}/{CENTERS}")[()].shape
            self._n_cells, self._n_dimensions = shape[0], shape[1]
            try:
                self._size_initial_cell = f.get(f"{CONST}/size_initial_cell")[()]
            except TypeError:
                if strict:
                    raise
                logger.warning("Could not load initial cell size.")

    def _clear_cache(self) -> None:
        self._write_times = None
        self._weights = None            # cell areas (2-D) / volumes (3-D)
        self._levels = None
        self._metric = None
        self._field_names = None
        self._vertices = None
        self._faces = None
        self._nodes = None

    # -- lazily loaded properties (names as in the reference: ``vertices`` are the cell CENTRES, ``nodes`` the corners)
    @property
    def write_times(self) -> List[str]:
        if self._write_times is None:
            with _h5file(self._path(), "r") as f:
                if DATA in f.keys():
                    self._write_times = list(f.get(DATA).keys())
        return self._write_times

    @property
    def weights(self) -> pt.Tensor:
        if self._weights is None:
            self._weights = (pow(self._size_initial_cell / pow(2, self.levels), self._n_dimensions)).squeeze()
        return self._weights

    @property
    def vertices(self) -> pt.Tensor:
        if self._vertices is None:
            self._vertices = pt.from_numpy(self._read(f"{GRID}/{CENTERS}"))
        return self._vertices

    @property
    def nodes(self) -> pt.Tensor:
        if self._nodes is None:
            self._nodes = pt.from_numpy(self._read(f"{GRID}/{VERTICES}"))
        return self._nodes

    @property
    def faces(self) -> pt.Tensor:
        if self._faces is None:
            self._faces = pt.from_numpy(self._read(f"{GRID}/{FACES}"))
        return self._faces

    @property
    def field_names(self) -> dict:
        if self._field_names is None:
            with _h5file(self._path(), "r") as f:
                self._field_names = {t: [n.split("_")[0] for n in f[f"{DATA}/{t}"].keys() if n.endswith("center")]
                                     for t in f[DATA].keys()}
        return self._field_names

    @property
    def levels(self) -> pt.Tensor:
        if self._levels is None:
            self._levels = pt.from_numpy(self._read(f"{CONST}/levels")).squeeze()
        return self._levels

    @property
    def metric(self) -> pt.Tensor:
        if self._metric is None:
            self._metric = pt.from_numpy(self._read(f"{CONST}/metric")).squeeze()
        return self._metric

    @property
    def load_path(self) -> str:
        return self._load_path

    @load_path.setter
    def load_path(self, value: str) -> None:
        self._load_path = value
        self._read_header(strict=True)
        self._clear_cache()

    @property
    def file_name(self) -> str:
        return self._file_name

    @file_name.setter
    def file_name(self, value: str) -> None:
        self._file_name = value
        self._read_header(strict=True)
        self._clear_cache()

    def load_snapshot(self, field_name: Union[List[str], str],
                      write_times: Union[str, List[str]] = None) -> Union[List[pt.Tensor], pt.Tensor]:
        """data matrix ``[N_cells, (N_dims,) N_times]`` of one field, or a list of them for several fields"""
        times = self.write_times if write_times is None else write_times
        times = [times] if isinstance(times, str) else times
        names = [field_name] if isinstance(field_name, str) else field_name
        matrices = []
        with _h5file(self._path(), "r") as f:
            for name in names:
                first = f.get(f"{DATA}/{times[0]}/{name}_center")[()]
                dm = pt.zeros(tuple(first.shape) + (len(times),), dtype=self._dtype)
                for i, t in enumerate(times):
                    dm[..., i] = pt.from_numpy(f.get(f"{DATA}/{t}/{name}_center")[()])
                matrices.append(dm)
        return matrices[0] if len(matrices) == 1 else matrices


class Datawriter:
    def __init__(self, file_path: str, file_name: str, mode: str = "w", mixed: bool = False):
        self._file_name = file_name
        self._mode = mode
        self._mixed = mixed
        self._file_path = file_path
        self._file = _h5file(join(self._file_path, self._file_name), self._mode)
        keys = self._file.keys()
        self._data = self._file[DATA] if DATA in keys else None
        self._const = self._file[CONST] if CONST in keys else None
        self._grid = self._file[GRID] if GRID in keys else None
        self._n_cells = None

    def close(self) -> None:
        self._file.close()

    def write_grid(self, loader: Dataloader) -> None:
        self._n_cells = loader.vertices.shape[0]
        self.write_data("centers", group="grid", data=loader.vertices)
        self.write_data("vertices", group="grid", data=loader.nodes)
        self.write_data("faces", group="grid", data=loader.faces)

    def write_data(self, name: str, data: any, group: str = "constant",
                   time_step: Union[int, float, str] = None) -> None:
        """one dataset in ``grid/``, ``constant/`` or ``data/<time_step>/`` (reference data.py:361-430)"""
        data = _np(data)
        if group == DATA and time_step is None:
            logger.warning(f"No time step for group 'data' provided. Writing data to the zeroth time step '{DATA}/0'.")
            time_step = "0"

        if time_step is not None or group == DATA:
            # with a known grid (write_grid was used) un-suffixed names get the location suffix the loader expects
            if self._n_cells is not None and not (name.endswith("center") or name.endswith("vertices")):
                name = f"{name}_center" if data.shape[0] == self._n_cells else f"{name}_vertices"
            if self._data is None or str(time_step) not in self._file[DATA].keys():
                self._data = self._file.create_group(f"{DATA}/{time_step}")
            else:
                self._data = self._file[f"{DATA}/{time_step}"]
            try:
                self._data.create_dataset(name, data=data)
            except ValueError:
                logger.warning(f"Field {name} already exists in the HDF file. Skipping field {name}.")
        elif group == CONST:
            self._const = self._file.create_group(CONST) if self._const is None else self._file[CONST]
            try:
                self._const.create_dataset(name, data=data)
            except ValueError:
                logger.warning(f"Field {name} already exists in time step {time_step}. Skipping field {name}.")
        elif group == GRID:
            self._grid = self._file.create_group(GRID) if self._grid is None else self._file[GRID]
            self._grid.create_dataset(name, data=data)
        else:
            logger.critical(f"Unknown group type, available types are '{DATA}', '{CONST}' and '{GRID}'.")
            exit()

    def write_xdmf_file(self) -> None:
        if not isfile(join(self._file_path, self._file_name)):
            logger.error(f"Could not find {join(self._file_path, self._file_name)}. Make sure the file exists and the "
                         f"provided path is correct.")
            exit(0)
        logger.info(f"Writing XDMF file for file {self._file_name}")
        XDMFWriter(self._file_path, self._file_name, mixed=self._mixed).write_xdmf()
        self.close()

    @property
    def mode(self) -> str:
        return self._mode

    @mode.setter
    def mode(self, value) -> None:
        self._mode = value
        self._file = _h5file(join(self._file_path, self._file_name), self._mode)

    @property
    def file_name(self) -> str:
        return self._file_name

    @property
    def n_cells(self) -> Union[int, None]:
        return self._n_cells

    @n_cells.setter
    def n_cells(self, value: int) -> None:
        self._n_cells = value


class XDMFWriter:
    """XDMF2 description of an S^3 HDF5 file (text identical to the reference's writer, data.py:504-777)"""

    _HEADER = '<?xml version="1.0"?>\n<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>\n<Xdmf Version="2.0">\n'

    def __init__(self, file_path: str, file_name: str, grid_name: str = "grid_s_cube", mixed: bool = False):
        self._file_path = file_path
        self._grid_name = grid_name
        self._mixed = mixed
        self._hdf_file_name = file_name
        self._file = _h5file(join(self._file_path, self._hdf_file_name), "r")
        self._xdmf_file_name = f"{self._hdf_file_name.split('.h5')[0]}.xdmf"
        self._check_grid()
        centers = self._file.get(f"{GRID}/{CENTERS}")[()]
        self._n_dimensions, self._n_cells = centers.shape[-1], centers.shape[0]
        self._n_faces = self._file.get(f"{GRID}/{FACES}")[()].shape[0]
        self._n_vertices = self._file.get(f"{GRID}/{VERTICES}")[()].shape[0]
        self._grid_type = "Mixed" if mixed else ("Quadrilateral" if self._n_dimensions == 2 else "Hexahedron")
        self._dims = "XY" if self._n_dimensions == 2 else "XYZ"
        self._keys_const_attributes = []

    # -- building blocks -------------------------------------------------------------------------------------------
    def _topology_and_geometry(self) -> str:
        face_dims = f"{self._n_faces}" if self._mixed else f"{self._n_faces} {pow(2, self._n_dimensions)}"
        return (f'<Topology TopologyType="{self._grid_type}" NumberOfElements="{self._n_faces}">\n'
                f'<DataItem Format="HDF" DataType="Int" Dimensions="{face_dims}">\n'
                f"{self._hdf_file_name}:/{GRID}/{FACES}\n"
                f'</DataItem>\n</Topology>\n<Geometry GeometryType="{self._dims}">\n'
                f'<DataItem Rank="2" Dimensions="{self._n_vertices} {self._n_dimensions}" '
                f'NumberType="Float" Precision="8" Format="HDF">\n'
                f"{self._hdf_file_name}:/{GRID}/{VERTICES}\n</DataItem>\n</Geometry>\n")

    def _attribute(self, name: str, h5_key: str) -> str:
        """cell- or node-centred attribute, or '' (with a warning) when the size matches neither"""
        shape = self._file.get(h5_key)[()].shape
        second = 1 if len(shape) == 1 else shape[1]
        if shape[0] == self._n_cells:
            center, n = "Cell", self._n_cells
        elif shape[0] == self._n_vertices:
            center, n = "Node", self._n_vertices
        else:
            logger.warning(f"Field in '{h5_key}' with a size of {shape} doesn't match the number of cells with "
                           f"N_cells = {self._n_cells} or the number of vertices with N_vertices = "
                           f"{self._n_vertices}. Skipping this field.")
            return ""
        return (f'<Attribute Name="{name}" AttributeType="Vector" Center="{center}">\n<DataItem '
                f'NumberType="Float" Precision="8" Format="HDF" Dimensions="{n} {second}">\n'
                f"{self._hdf_file_name}:/{h5_key}\n</DataItem>\n</Attribute>\n")

    def _write_attributes(self) -> str:
        return "".join(self._attribute(k, f"{CONST}/{k}") for k in self._keys_const_attributes)

    # -- writers -------------------------------------------------------------------------------------------------
    def write_xdmf(self) -> None:
        self._keys_const_attributes = self._get_const_keys()
        if self._check_data():
            self._write_temporal_grid()
        else:
            self._write_const_grid()

    def _write_temporal_grid(self) -> None:
        with open(join(self._file_path, self._xdmf_file_name), "w") as f_out:
            f_out.write(self._HEADER)
            f_out.write(f'<Domain>\n<Grid Name="{self._grid_name}" GridType="Collection" CollectionType="temporal">\n')
            for i, t in enumerate(sorted(self._file.get(DATA).keys(), key=lambda x: float(x))):
                f_out.write(f'<Grid Name="{self._grid_name} {t}" GridType="Uniform">\n<Time Value="{t}"/>\n')
                f_out.write(self._topology_and_geometry())
                if i == 0:                          # constant fields go into the first time step
                    f_out.write(self._write_attributes())
                for k in self._file[f"{DATA}/{t}"].keys():
                    parts = k.split("_")           # <field name>_<center|vertices>
                    f_out.write(self._attribute("_".join(parts[:-1]) if len(parts) > 1 else k, f"{DATA}/{t}/{k}"))
                f_out.write('</Grid>\n')
            f_out.write('</Grid>\n</Domain>\n</Xdmf>')

    def _write_const_grid(self) -> None:
        with open(join(self._file_path, self._xdmf_file_name), "w") as f_out:
            f_out.write(self._HEADER)
            f_out.write(f'<Domain>\n<Grid Name="{self._grid_name}" GridType="Uniform">\n')
            f_out.write(self._topology_and_geometry())
            f_out.write(self._write_attributes())
            f_out.write("</Grid>\n</Domain>\n</Xdmf>")

    def _get_const_keys(self) -> list:
        if CONST not in self._file.keys():
            logger.info("Couldn't find any constant fields to write.")
            return []
        keys = []
        for k in self._file[CONST].keys():
            shape = self._file.get(f"{CONST}/{k}")[()].shape
            if shape and shape[0] in (self._n_cells, self._n_vertices):
                keys.append(k)
        return keys

    def _check_data(self) -> bool:
        return DATA in self._file.keys()

    def _check_grid(self) -> None:
        if GRID not in self._file.keys():
            logger.error("Found no grid in the provided HDF5 file. Unable to create XDMF file without a grid.")
            exit(0)
        for key, what in ((FACES, "faces"), (CENTERS, "centers"), (VERTICES, "vertices")):
            if key not in self._file[GRID].keys():
                logger.error(f"Unable to find cell {what} in group {GRID}. Make sure the key to the cell {what} is "
                             f"present and named {key}.")
                exit(0)
