"""ctypes binding of libflacarray_hip.so (C ABI: include/flacarray_hip.h).

There is no CPU fallback: if the library cannot be loaded, or no HIP device is visible when a
compute entry point is called, the call fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLACARRAY_HIP_LIB", os.path.join(_HERE, "lib", "libflacarray_hip.so"))

# every symbol include/flacarray_hip.h declares
SYMBOLS = [
    "encode_i32",
    "encode_i32_threaded",
    "decode_i32",
    "decode_i64",
    "encode_i64",
    "encode_i64_threaded",
    "float64_to_int64",
    "int64_to_float64",
    "float32_to_int32",
    "int32_to_float32",
    "fa_encode_workspace_bytes",
    "fa_encode_i32_device_begin",
    "fa_encode_i32_device_finish",
    "fa_encode_single_pass_supported",
    "fa_encode_capacity_bytes",
    "fa_encode_single_pass_workspace_bytes",
    "fa_encode_i32_device",
    "fa_encode_f32_device",
    "fa_decode_i32_device",
    "fa_decode_slices_i32_device",
    "fa_decode_i64_device",
    "fa_encode_workspace_bytes_i64",
    "fa_encode_i64_device_begin",
    "fa_encode_i64_device_finish",
    "fa_encode_capacity_bytes_i64",
    "fa_encode_single_pass_workspace_bytes_i64",
    "fa_encode_i64_device",
    "fa_float64_to_int64_device",
    "fa_int64_to_float64_device",
    "fa_decode_slices_i64_device",
    "fa_float32_to_int32_device",
    "fa_int32_to_float32_device",
    "fa_decode_index_create",
    "fa_decode_index_destroy",
    "fa_decode_indexed",
    "fa_set_decode_verify",
    "fa_encode_f32_host",
    "fa_encode_f64_host",
    "fa_decode_f32_host",
    "fa_decode_f64_host",
    "fa_decode_indexed_host",
    "fa_pinned_alloc",
    "fa_pinned_free",
    "fa_profile_enable",
    "fa_profile_last",
    "fa_profile_read",
    "fa_release_scratch",
    "fa_device_count",
    "fa_version",
    "fa_abi_version",
]

ABI_VERSION = 2  # FA_ABI_VERSION of include/flacarray_hip.h this binding was written against

# error bits (flacarray.h:20-40 + this library's additions)
ERROR_DEVICE = 1 << 24
ERROR_NAN_INPUT = 1 << 25

_lib = None
_libc = None


def lib():
    """Load (once) and return the shared library, with argument types declared."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m flacarray_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback."
        )
    # If PyTorch-ROCm is present it must load ITS bundled HIP runtime first: the library then
    # binds to the runtime already in the process (same soname) instead of bringing a second
    # one, whose pointers and streams torch tensors would not belong to.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    try:
        L.fa_abi_version.restype = ctypes.c_int
        found = L.fa_abi_version()
    except AttributeError:
        found = None
    if found != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI revision {found}, this binding needs {ABI_VERSION}: rebuild it (python -m flacarray_amd.build --force)")
    i64, u32, vp, cint = ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int
    pi64 = ctypes.POINTER(ctypes.c_int64)
    L.encode_i32.argtypes = [vp, i64, i64, u32, pi64, vp, ctypes.POINTER(vp)]
    L.encode_i32.restype = cint
    L.encode_i32_threaded.argtypes = L.encode_i32.argtypes
    L.encode_i32_threaded.restype = cint
    L.decode_i32.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp, ctypes.c_bool]
    L.decode_i32.restype = cint
    L.float32_to_int32.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    L.float32_to_int32.restype = cint
    L.int32_to_float32.argtypes = [vp, i64, i64, vp, vp, vp]
    L.int32_to_float32.restype = None
    L.fa_encode_workspace_bytes.argtypes = [i64, i64, u32]
    L.fa_encode_workspace_bytes.restype = i64
    L.fa_encode_i32_device_begin.argtypes = [vp, i64, i64, u32, vp, i64, vp, vp, pi64, vp, vp]
    L.fa_encode_i32_device_begin.restype = cint
    L.fa_encode_i32_device_finish.argtypes = [i64, i64, u32, vp, vp, vp, vp]
    L.fa_encode_i32_device_finish.restype = cint
    L.fa_encode_single_pass_supported.argtypes = [i64, i64, u32]
    L.fa_encode_single_pass_supported.restype = cint
    L.fa_encode_capacity_bytes.argtypes = [i64, i64, u32]
    L.fa_encode_capacity_bytes.restype = i64
    L.fa_encode_single_pass_workspace_bytes.argtypes = [i64, i64, u32]
    L.fa_encode_single_pass_workspace_bytes.restype = i64
    L.fa_encode_i32_device.argtypes = [vp, i64, i64, u32, vp, i64, vp, i64, vp, vp, pi64, vp, vp]
    L.fa_encode_i32_device.restype = cint
    if hasattr(L, "fa_encode_i64_device") or not os.environ.get("FLACARRAY_HIP_LIB"):  # (an older diagnostic build may lack them)
        L.fa_encode_i64_device.argtypes = L.fa_encode_i32_device.argtypes
        L.fa_encode_i64_device.restype = cint
        for name in ("fa_encode_capacity_bytes_i64", "fa_encode_single_pass_workspace_bytes_i64"):
            getattr(L, name).argtypes = [i64, i64, u32]
            getattr(L, name).restype = i64
    L.fa_encode_f32_device.argtypes = [vp, i64, i64, u32, vp, vp, i64, vp, i64, vp, vp, vp, vp, pi64, vp, vp]
    L.fa_encode_f32_device.restype = cint
    L.fa_decode_i32_device.argtypes = [vp, i64, vp, vp, i64, i64, i64, i64, vp, vp, vp, vp, vp, cint]
    L.fa_decode_i32_device.restype = cint
    L.fa_decode_slices_i32_device.argtypes = [vp, i64, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, cint]
    L.fa_decode_slices_i32_device.restype = cint
    L.fa_float32_to_int32_device.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp]
    L.fa_float32_to_int32_device.restype = cint
    L.fa_int32_to_float32_device.argtypes = [vp, i64, i64, vp, vp, vp, vp]
    L.fa_int32_to_float32_device.restype = cint
    L.fa_decode_index_create.argtypes = [vp, i64, vp, vp, i64, i64, cint, ctypes.POINTER(vp), vp]
    L.fa_decode_index_create.restype = cint
    L.fa_decode_index_destroy.argtypes = [vp]
    L.fa_decode_index_destroy.restype = None
    L.fa_decode_indexed.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, cint]
    L.fa_decode_indexed.restype = cint
    L.fa_encode_f32_host.argtypes = [vp, i64, i64, u32, vp, pi64, vp, ctypes.POINTER(vp), vp, vp]
    L.fa_encode_f32_host.restype = cint
    L.fa_encode_f64_host.argtypes = [vp, i64, i64, u32, vp, pi64, vp, ctypes.POINTER(vp), vp, vp]
    L.fa_encode_f64_host.restype = cint
    L.fa_decode_f32_host.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp, vp, vp]
    L.fa_decode_f32_host.restype = cint
    L.fa_decode_f64_host.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp, vp, vp]
    L.fa_decode_f64_host.restype = cint
    L.fa_decode_indexed_host.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, cint]
    L.fa_decode_indexed_host.restype = cint
    L.fa_pinned_alloc.argtypes = [i64]
    L.fa_pinned_alloc.restype = vp
    L.fa_pinned_free.argtypes = [vp]
    L.fa_pinned_free.restype = None
    L.fa_set_decode_verify.argtypes = [cint]
    L.fa_set_decode_verify.restype = cint
    L.fa_profile_enable.argtypes = [cint]
    L.fa_profile_enable.restype = None
    L.fa_profile_last.argtypes = [ctypes.POINTER(ctypes.c_float)]
    L.fa_profile_last.restype = cint
    L.fa_profile_read.argtypes = [ctypes.POINTER(ctypes.c_float), cint]
    L.fa_profile_read.restype = cint
    L.fa_release_scratch.argtypes = []
    L.fa_release_scratch.restype = None
    L.fa_device_count.argtypes = []
    L.fa_device_count.restype = cint
    L.fa_version.argtypes = []
    L.fa_version.restype = ctypes.c_char_p
    # two-channel (int64 / float64) entry points; an older diagnostic build selected through
    # FLACARRAY_HIP_LIB for an A/B run may lack them
    if hasattr(L, "decode_i64") or not os.environ.get("FLACARRAY_HIP_LIB"):
        _declare_i64(L)
    _lib = L
    return L


def _declare_i64(L):
    cint = ctypes.c_int
    L.decode_i64.argtypes = L.decode_i32.argtypes
    L.decode_i64.restype = cint
    for name in ("encode_i64", "encode_i64_threaded"):
        getattr(L, name).argtypes = L.encode_i32.argtypes
        getattr(L, name).restype = cint
    L.float64_to_int64.argtypes = L.float32_to_int32.argtypes
    L.float64_to_int64.restype = cint
    L.int64_to_float64.argtypes = L.int32_to_float32.argtypes
    L.int64_to_float64.restype = None
    L.fa_float64_to_int64_device.argtypes = L.fa_float32_to_int32_device.argtypes
    L.fa_float64_to_int64_device.restype = cint
    L.fa_int64_to_float64_device.argtypes = L.fa_int32_to_float32_device.argtypes
    L.fa_int64_to_float64_device.restype = cint
    L.fa_encode_workspace_bytes_i64.argtypes = L.fa_encode_workspace_bytes.argtypes
    L.fa_encode_workspace_bytes_i64.restype = ctypes.c_int64
    L.fa_encode_i64_device_begin.argtypes = L.fa_encode_i32_device_begin.argtypes
    L.fa_encode_i64_device_begin.restype = cint
    L.fa_encode_i64_device_finish.argtypes = L.fa_encode_i32_device_finish.argtypes
    L.fa_encode_i64_device_finish.restype = cint
    L.fa_decode_i64_device.argtypes = L.fa_decode_i32_device.argtypes
    L.fa_decode_i64_device.restype = cint
    L.fa_decode_slices_i64_device.argtypes = L.fa_decode_slices_i32_device.argtypes
    L.fa_decode_slices_i64_device.restype = cint


def libc_free(addr):
    global _libc
    if _libc is None:
        _libc = ctypes.CDLL(None)
        _libc.free.argtypes = [ctypes.c_void_p]
        _libc.free.restype = None
    _libc.free(addr)


def libc_madvise(addr, length, advice):
    """madvise(2) on a range of this process' memory; returns its result (0 = accepted)."""
    global _libc
    if _libc is None:
        libc_free(None)  # (loads libc; free(NULL) does nothing)
    _libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    _libc.madvise.restype = ctypes.c_int
    return _libc.madvise(ctypes.c_void_p(addr), length, advice)


def require_device():
    """Raise unless a HIP device is visible (the product path never falls back to the CPU)."""
    if lib().fa_device_count() <= 0:
        raise RuntimeError("flacarray_amd: no HIP device visible; the MI355X path has no CPU fallback")
