"""A test double for the small part of h5py that quflow_amd.simulation.H5Store uses (h5py is not installable in the
build container, so the real backend's two tests skip there).  It executes H5Store's own logic -- group and dataset
paths, the growing first axis, attribute round trips incl. pickled bytes -- against a file that is a pickle of a dict.
It says nothing about the HDF5 format: interoperability with files written by the reference stays unverified."""
import os
import pickle

import numpy as np


class _Attrs(dict):
    def __setitem__(self, key, value):
        dict.__setitem__(self, key, np.array(value) if isinstance(value, (list, tuple)) else value)


class Dataset:
    def __init__(self, shape, dtype, maxshape=None, chunks=None):
        self._a = np.zeros(shape, dtype=dtype)
        self.maxshape = maxshape
        self.chunks = chunks

    @property
    def shape(self):
        return self._a.shape

    @property
    def dtype(self):
        return self._a.dtype

    def resize(self, size, axis=None):
        if self.maxshape is None or self.maxshape[axis] is not None:
            raise TypeError("only a dataset created with maxshape=(None, ...) grows along axis 0")
        new = np.zeros((size,) + self._a.shape[1:], dtype=self._a.dtype)
        n = min(size, self._a.shape[0])
        new[:n] = self._a[:n]
        self._a = new

    def __getitem__(self, idx):
        out = self._a[idx]
        return out.copy() if isinstance(out, np.ndarray) else out

    def __setitem__(self, idx, value):
        self._a[idx] = value


class Group:
    def __init__(self):
        self.items = {}
        self.attrs = _Attrs()

    def keys(self):
        return self.items.keys()


class File:
    """h5py.File(path, mode) as a context manager; paths like '/', '/args/', '/mat'."""

    def __init__(self, path, mode="r"):
        self.path, self.mode = path, mode
        if mode == "w":
            self.root = Group()
        else:
            if not os.path.isfile(path):
                raise OSError("unable to open file: %s" % path)
            with open(path, "rb") as f:
                self.root = pickle.load(f)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.mode in ("w", "r+", "a") and exc[0] is None:
            with open(self.path, "wb") as f:
                pickle.dump(self.root, f)
        return False

    def _walk(self, path, create_last=None):
        parts = [p for p in path.split("/") if p]
        node = self.root
        for i, p in enumerate(parts):
            if p not in node.items:
                if create_last is not None and i == len(parts) - 1:
                    node.items[p] = create_last
                else:
                    raise KeyError("object '%s' doesn't exist" % path)
            node = node.items[p]
        return node

    def __getitem__(self, path):
        return self._walk(path)

    def create_group(self, path):
        if self.mode == "r":
            raise ValueError("file is read-only")
        parts = [p for p in path.split("/") if p]
        node = self.root
        for p in parts:
            node = node.items.setdefault(p, Group())
        return node

    def create_dataset(self, path, shape, dtype=None, maxshape=None, chunks=None):
        if self.mode == "r":
            raise ValueError("file is read-only")
        parts = [p for p in path.split("/") if p]
        parent = self.root
        for p in parts[:-1]:
            parent = parent.items[p]
        if parts[-1] in parent.items:
            raise ValueError("name already exists: %s" % path)
        if chunks is not None and len(chunks) != len(shape):
            raise ValueError("chunks must have the dataset's rank")
        ds = Dataset(shape, dtype, maxshape, chunks)
        parent.items[parts[-1]] = ds
        return ds
