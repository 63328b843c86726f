"""Stand-in for `h5py` (HDF5 persistence; off the hot path, never called)."""


class File:  # noqa: D401 - placeholder
    pass


class Dataset:
    pass
