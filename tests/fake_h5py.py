"""Minimal in-memory stand-in for h5py (File / groups / datasets) so the HDF5 + XDMF host logic can be tested in an
image without h5py.  Test infrastructure only."""
import os
import sys
import types

import numpy as np

_STORE = {}


class _Dataset:
    def __init__(self, data):
        self._a = np.array(data)

    def __getitem__(self, key):
        return self._a if key == () else self._a[key]

    @property
    def shape(self):
        return self._a.shape


class _Group:
    def __init__(self):
        self._items = {}

    def keys(self):
        return self._items.keys()

    def __contains__(self, k):
        return k in self._items

    def _walk(self, path, create=False):
        node = self
        for part in [p for p in path.split("/") if p]:
            if part not in node._items:
                if not create:
                    return None
                node._items[part] = _Group()
            node = node._items[part]
        return node

    def __getitem__(self, path):
        node = self._walk(path)
        if node is None:
            raise KeyError(path)
        return node

    def get(self, path):
        return self._walk(path)

    def create_group(self, path):
        return self._walk(path, create=True)

    def create_dataset(self, name, data=None):
        if name in self._items:
            raise ValueError(f"dataset {name} exists")
        self._items[name] = _Dataset(data)
        return self._items[name]


class File(_Group):
    def __init__(self, path, mode="r"):
        super().__init__()
        self._path = os.path.abspath(path)
        if mode == "w" or self._path not in _STORE:
            if mode == "r":
                raise FileNotFoundError(path)
            _STORE[self._path] = {}
            open(self._path, "w").close()          # Datawriter.write_xdmf_file checks isfile()
        self._items = _STORE[self._path]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def install():
    mod = types.ModuleType("h5py")
    mod.File = File
    sys.modules["h5py"] = mod


def dump(path):
    """{dataset path: ndarray} of a fake file"""
    out = {}

    def rec(items, prefix):
        for k, v in items.items():
            if isinstance(v, _Group):
                rec(v._items, f"{prefix}{k}/")
            else:
                out[f"{prefix}{k}"] = v._a
    rec(_STORE[os.path.abspath(path)], "")
    return out
