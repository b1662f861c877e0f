"""A numpy-backed stand-in for the slice of `h5py` the A-NeRF dataset code uses -- TEST INFRASTRUCTURE (h5py is not in the
build image, and nothing here ships).

Two users:
  * tests/golden/gen_golden_dataset.py installs it as `sys.modules["h5py"]` so that the REFERENCE's own writer
    (`core/process_spin.py:234-297 write_to_h5py`) and dataset classes (`core/dataset.py:20-420`, `core/load_surreal.py:302`,
    `core/load_mixamo.py:161`) run in the build container and produce the dataset fixtures;
  * tests/test_dataset_layout.py installs it to take `a-nerf_amd/dataset.py`'s `.h5` branch (`_open` -> `h5py.File(path, "r")`)
    through the same reads.

What is modelled: `File(path, mode, swmr=...)` as a mapping name -> Dataset (`keys`, `in`, `[]`, `close`, context manager),
`File.create_dataset(name, shape, dtype, chunks=, compression=)`, and Dataset reads / writes `ds[:]`, `ds[()]`, `ds[i]`,
`ds[i, sorted_index_array]`, `ds[i] = row`, plus `shape`, `dtype`, `len()`.  As in HDF5, a fancy index must be strictly
increasing (h5py raises otherwise -- the reason the reference sorts its sampled pixels, dataset.py:325).
The container on disk is numpy's .npz, whatever the file is called.
"""
import numpy as np


class Dataset:
    def __init__(self, arr):
        self._a = arr

    shape = property(lambda self: self._a.shape)
    dtype = property(lambda self: self._a.dtype)
    ndim = property(lambda self: self._a.ndim)

    def __len__(self):
        return len(self._a)

    @staticmethod
    def _check(key):
        for k in key if isinstance(key, tuple) else (key,):
            if isinstance(k, (list, np.ndarray)) and np.asarray(k).dtype != bool:
                k = np.asarray(k)
                if k.ndim != 1 or (k.size > 1 and not np.all(np.diff(k) > 0)):
                    raise TypeError("Indexing elements must be in increasing order")      # h5py's message

    def __getitem__(self, key):
        self._check(key)
        out = self._a[key]
        return out.copy() if isinstance(out, np.ndarray) else out

    def __setitem__(self, key, value):
        self._check(key)
        self._a[key] = value

    def astype(self, dtype):
        return Dataset(self._a.astype(dtype))


class File:
    def __init__(self, path, mode="r", swmr=False, **_):
        self.filename, self.mode, self._d, self._open = str(path), mode, {}, True
        if mode in ("r", "r+", "a"):
            with np.load(self.filename, allow_pickle=False) as z:
                self._d = {k: Dataset(np.asarray(z[k])) for k in z.files}
        elif mode not in ("w", "x", "w-"):
            raise ValueError(f"h5shim: mode {mode!r}")

    def keys(self):
        return self._d.keys()

    def __iter__(self):
        return iter(self._d)

    def __contains__(self, k):
        return k in self._d

    def __getitem__(self, k):
        return self._d[k]

    def __len__(self):
        return len(self._d)

    def create_dataset(self, name, shape=None, dtype=None, data=None, chunks=None, compression=None, **_):
        if self.mode == "r":
            raise ValueError("h5shim: file is read-only")
        if data is not None:
            arr = np.array(data, dtype=dtype)
        else:
            arr = np.zeros(shape, dtype=np.dtype(dtype))
        if chunks is not None and len(chunks) != arr.ndim:
            raise ValueError("h5shim: chunk rank differs from dataset rank")                # h5py's check
        self._d[name] = Dataset(arr)
        return self._d[name]

    def close(self):
        if self._open and self.mode != "r":
            with open(self.filename, "wb") as fh:                                           # a file object: no ".npz" gets appended
                np.savez(fh, **{k: v._a for k, v in self._d.items()})
        self._open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def install():
    """make `import h5py` resolve to this module (generators / tests only)"""
    import sys
    me = sys.modules[__name__]
    sys.modules["h5py"] = me
    return me
