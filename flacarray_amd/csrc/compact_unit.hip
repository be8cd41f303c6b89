// compact_unit.hip -- second translation unit of libflacarray_hip.so: the compaction kernels (K5) and their launchers,
// compiled with the default scheduling strategy (see the note at the top of namespace fa in encode_kernels.hpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FA_UNIT_COMPACT 1
#define FA_HAVE_K5_LAUNCHERS 1
#include "encode_kernels.hpp"
