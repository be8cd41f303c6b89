// fused_unit.hip -- third translation unit of libflacarray_hip.so: the single-pass encoder K3F (encode_fused.hpp), its
// stream-header kernel and their launchers, compiled with the default scheduling strategy (see build.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FA_UNIT_FUSED 1
#define FA_SPLIT_UNITS 1
#include "encode_fused.hpp"
