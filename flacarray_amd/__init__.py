"""flacarray_amd -- MI355X-native FLAC encode/decode path behind flacarray's operator API.

Drop-in names of the reference (hpc4cmb/flacarray): `array_compress`, `array_decompress`,
`array_decompress_slice`, `encode_flac`, `decode_flac`, `float_to_int`, `int_to_float`,
`FlacArray`.  `array_encode` / `array_decode` are aliases for the spelling used in
BASELINE.json.  All compute runs in hand-written HIP kernels (libflacarray_hip.so); there is
no CPU fallback.
"""
from .array import FlacArray
from .compress import array_compress
from .decompress import array_decompress, array_decompress_slice
from .libflacarray import (
    DeviceDecodeIndex,
    decode_flac,
    decode_flac_device,
    decode_slices_device,
    encode_flac,
    encode_flac_device,
    encode_flac_device_f32,
    float32_to_int32_device,
    set_decode_verify,
)
from .utils import float_to_int, int_to_float, keep_select

array_encode = array_compress
array_decode = array_decompress

__version__ = "0.1.0"

__all__ = [
    "FlacArray",
    "DeviceDecodeIndex",
    "array_compress",
    "array_decompress",
    "array_decompress_slice",
    "array_encode",
    "array_decode",
    "encode_flac",
    "decode_flac",
    "encode_flac_device",
    "encode_flac_device_f32",
    "decode_flac_device",
    "decode_slices_device",
    "float32_to_int32_device",
    "set_decode_verify",
    "float_to_int",
    "int_to_float",
    "keep_select",
]
