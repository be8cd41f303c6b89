"""Zarr layout, format version 1 (reference: src/flacarray/zarr.py:145-447).

The Zarr schema is the HDF5 one name for name (datasets `compressed`, `stream_starts`, `stream_bytes`,
optional `stream_offsets` / `stream_gains`; attributes `flacarray_format_version`, `flac_channels`,
`stream_size`), so the four entry points are the functions of flacarray_amd.hdf5, which create datasets through
`create_array` when the group is a Zarr group (hdf5.py:_create).  Single process; the reference's `ZarrGroup`
context manager and the MPI write path are outside this hot path.
"""
from .hdf5 import read_array, read_compressed, write_array, write_compressed

__all__ = ["write_array", "read_array", "write_compressed", "read_compressed"]
