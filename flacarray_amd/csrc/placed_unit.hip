// placed_unit.hip -- fourth translation unit of libflacarray_hip.so: the placing encoder K3G (encode_placed.hpp: K3's
// frame body in a persistent ticket loop, frames moved to their final offsets by the waves that packed them), compiled
// like the slot encoder it shares its frame body with (max-ILP scheduling, two waves per SIMD: see build.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FA_UNIT_PLACED 1
#define FA_SPLIT_UNITS 1
#include "encode_placed.hpp"
