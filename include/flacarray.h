/*
 * flacarray.h -- stand-in for the reference's header of the same name, for builds that link the
 * reference's Cython binding against libflacarray_hip.so instead of libFLAC.
 *
 * The reference's binding says `cdef extern from "flacarray.h"` (src/flacarray/libflacarray/
 * libflacarray.pyx:18) and the reference's own flacarray.h pulls in <FLAC/stream_encoder.h> and
 * <FLAC/stream_decoder.h> (flacarray.h:14-15).  Put THIS file on the include path instead and the
 * generated C of libflacarray.pyx compiles without any libFLAC header: it needs nothing but the ten
 * prototypes below (libflacarray.pyx:19-110), which include/flacarray_hip.h declares with the
 * reference's exact signatures (flacarray.h:209-311).
 *
 * Not declared here, on purpose: ArrayUint8 and the encoder / decoder callback structures
 * (flacarray.h:44-141).  They are internals of compress.c / decompress.c, which this library replaces.
 */
#ifndef FLACARRAY_H_HIP_STANDIN
#define FLACARRAY_H_HIP_STANDIN

#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "flacarray_hip.h"

/* The reference's unprefixed error names (flacarray.h:20-40), same bit values.  Codes the GPU path can
 * never raise (libFLAC setter failures, ERROR_ENCODE_FINISH, ...) keep their values so that code which
 * tests for them still compiles. */
#define ERROR_NONE FA_ERROR_NONE
#define ERROR_ALLOC FA_ERROR_ALLOC
#define ERROR_INVALID_LEVEL FA_ERROR_INVALID_LEVEL
#define ERROR_ZERO_NSTREAM FA_ERROR_ZERO_NSTREAM
#define ERROR_ZERO_STREAMSIZE FA_ERROR_ZERO_STREAMSIZE
#define ERROR_ENCODE_SET_COMP_LEVEL (1 << 4)
#define ERROR_ENCODE_SET_BLOCK_SIZE (1 << 5)
#define ERROR_ENCODE_SET_CHANNELS (1 << 6)
#define ERROR_ENCODE_SET_BPS (1 << 7)
#define ERROR_ENCODE_INIT FA_ERROR_ENCODE_INIT
#define ERROR_ENCODE_PROCESS FA_ERROR_ENCODE_PROCESS
#define ERROR_ENCODE_FINISH (1 << 10)
#define ERROR_ENCODE_COLLECT (1 << 11)
#define ERROR_DECODE_READ_ZEROBUF (1 << 12)
#define ERROR_DECODE_INIT FA_ERROR_DECODE_INIT
#define ERROR_DECODE_PROCESS FA_ERROR_DECODE_PROCESS
#define ERROR_DECODE_FINISH (1 << 15)
#define ERROR_DECODE_STREAMSIZE FA_ERROR_DECODE_STREAMSIZE
#define ERROR_DECODE_SAMPLE_RANGE FA_ERROR_DECODE_SAMPLE_RANGE
#define ERROR_DECODE_SEEK FA_ERROR_DECODE_SEEK
#define ERROR_CONVERT_TYPE FA_ERROR_CONVERT_TYPE

#endif /* FLACARRAY_H_HIP_STANDIN */
