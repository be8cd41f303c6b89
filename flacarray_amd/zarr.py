"""Zarr "format version 1" layout of a compressed array (reference: src/flacarray/zarr.py:160-330
write side, src/flacarray/zarr_load_v1.py read side).

The schema is the HDF5 one under the same names (`zarr_load_v1.zarr_names` == `hdf5_utils.hdf5_names`):
group attrs `flacarray_format_version`, `flacarray_software_version`, `flac_channels`; arrays
`stream_starts` (attr `stream_size`), `stream_bytes`, optional `stream_offsets` / `stream_gains`,
`compressed`.  Only array creation differs between h5py, zarr-2 and zarr-3 groups, which
flacarray_amd.hdf5._create handles, so these are the same functions under the reference's names.
Single-process I/O (see flacarray_amd.hdf5); zarr itself is only needed to open stores.
"""
from .hdf5 import hdf5_names as zarr_names  # noqa: F401
from .hdf5 import read_array, read_compressed, write_array, write_compressed  # noqa: F401
