/*
 * flacarray_hip.h -- C ABI of the MI355X-native FLAC encode/decode path.
 *
 * The first group of entry points has exactly the names, argument lists, return codes and
 * ownership rules of the reference's C layer, so the reference's Cython binding
 * (src/flacarray/libflacarray/libflacarray.pyx:18-110 `cdef extern from "flacarray.h"`) can be
 * linked against libflacarray_hip.so instead of compress.c/decompress.c/utils.c + libFLAC.
 * The second group takes DEVICE pointers (data already resident in HBM) and is what the
 * Python layer, the tests and bench.py use.
 *
 * No torch / HIP types appear in any signature: `void* stream` is a hipStream_t passed as an
 * opaque pointer (NULL = default stream).
 */
#ifndef FLACARRAY_HIP_H
#define FLACARRAY_HIP_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error bit-codes, identical to the reference (src/flacarray/libflacarray/flacarray.h:20-40). */
#define FA_ERROR_NONE 0
#define FA_ERROR_ALLOC (1 << 0)
#define FA_ERROR_INVALID_LEVEL (1 << 1)
#define FA_ERROR_ZERO_NSTREAM (1 << 2)
#define FA_ERROR_ZERO_STREAMSIZE (1 << 3)
#define FA_ERROR_ENCODE_INIT (1 << 8)
#define FA_ERROR_ENCODE_PROCESS (1 << 9)
#define FA_ERROR_DECODE_INIT (1 << 13)
#define FA_ERROR_DECODE_PROCESS (1 << 14)
#define FA_ERROR_DECODE_STREAMSIZE (1 << 16)
#define FA_ERROR_DECODE_SAMPLE_RANGE (1 << 17)
#define FA_ERROR_DECODE_SEEK (1 << 18)
#define FA_ERROR_CONVERT_TYPE (1 << 19)
/* additions of this library */
#define FA_ERROR_DEVICE (1 << 24)   /* a HIP runtime call failed (no GPU, out of memory, ...) */
#define FA_ERROR_NAN_INPUT (1 << 25) /* float32_to_int32 saw a NaN */

/* ---------------------------------------------------------------------------------------
 * Group 1: host-pointer drop-ins for the reference C ABI
 * ------------------------------------------------------------------------------------- */

/* replaces encode_i32, flacarray.h:209-217 (compress.c:440).  `*bytes` is malloc()'d by the
 * callee and owned by the caller (free()), as in compress.c:251,414. */
int encode_i32(int32_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
               int64_t* starts, unsigned char** bytes);

/* replaces encode_i32_threaded, flacarray.h:219-227 (compress.c:461); same work on the GPU */
int encode_i32_threaded(int32_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
                        int64_t* starts, unsigned char** bytes);

/* replaces decode_i32, flacarray.h:249-259 (decompress.c:318).  first_sample/last_sample < 0
 * decodes whole streams; otherwise the half-open range [first_sample, last_sample) of every
 * stream, row stride last_sample-first_sample (decompress.c:209-222,254). */
int decode_i32(unsigned char* const bytes, int64_t* const starts, int64_t* const nbytes, int64_t n_stream,
               int64_t stream_size, int64_t first_sample, int64_t last_sample, int32_t* data, bool use_threads);

/* replace encode_i64 / encode_i64_threaded, flacarray.h:229-247 (compress.c:482-540): int64 samples
 * as two-channel 32-bit streams, channel 0 = low word, channel 1 = high word (utils.c:96-123); the
 * channels are coded independently (channel assignment "left/right") */
int encode_i64(int64_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
               int64_t* starts, unsigned char** bytes);
int encode_i64_threaded(int64_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
                        int64_t* starts, unsigned char** bytes);

/* replaces decode_i64, flacarray.h:261-271 (decompress.c:343-375): two-channel streams, int64
 * sample = (channel 1 << 32) | (channel 0 as unsigned) (utils.c:96-123).  Reads streams with any
 * stereo channel assignment (left/right, left/side, side/right, mid/side). */
int decode_i64(unsigned char* const bytes, int64_t* const starts, int64_t* const nbytes, int64_t n_stream,
               int64_t stream_size, int64_t first_sample, int64_t last_sample, int64_t* data, bool use_threads);

/* replaces float32_to_int32, flacarray.h:275-283 (utils.c:160); quanta == NULL: per-stream
 * quanta from the data range */
int float32_to_int32(float const* input, int64_t n_stream, int64_t stream_size, float const* quanta, int32_t* output,
                     float* offsets, float* gains);

/* replaces int32_to_float32, flacarray.h:304-311 (utils.c:350) */
void int32_to_float32(int32_t const* input, int64_t n_stream, int64_t stream_size, float const* offsets,
                      float const* gains, float* output);

/* replace float64_to_int64 / int64_to_float64, flacarray.h:285-302 (utils.c:245-348) */
int float64_to_int64(double const* input, int64_t n_stream, int64_t stream_size, double const* quanta, int64_t* output,
                     double* offsets, double* gains);
void int64_to_float64(int64_t const* input, int64_t n_stream, int64_t stream_size, double const* offsets,
                      double const* gains, double* output);

/* ---------------------------------------------------------------------------------------
 * Group 2: device-pointer entry points (every pointer named d_* is HBM memory)
 * ------------------------------------------------------------------------------------- */

/* bytes of scratch fa_encode_i32_device_begin needs for this problem size */
int64_t fa_encode_workspace_bytes(int64_t n_stream, int64_t stream_size, uint32_t level);

/* Encode, phase 1: analyse and bit-pack every frame into the workspace, size the output.
 * Writes d_starts[n_stream], d_nbytes[n_stream] (device) and *h_total_bytes (host; the call
 * synchronises the stream to deliver it).  d_info may be NULL; otherwise it receives 8 int32
 * per frame {type, order, partition order, wasted bits, shift, precision, bytes, blocksize}. */
int fa_encode_i32_device_begin(const int32_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level,
                               void* d_workspace, int64_t workspace_bytes, int64_t* d_starts, int64_t* d_nbytes,
                               int64_t* h_total_bytes, int32_t* d_info, void* stream);

/* Encode, phase 2: assemble the blob (stream headers, byte-exact concatenation, CRC-16) into
 * d_bytes[*h_total_bytes].  Same arguments as phase 1. */
int fa_encode_i32_device_finish(int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                                const int64_t* d_starts, unsigned char* d_bytes, void* stream);

/* Single-pass encode: ONE kernel analyses every frame, sizes it, obtains its byte offset from a look-back over the frames
 * before it (a scanner wave turns published sizes into offsets) and the frame ends up -- CRC-16 included -- at its final
 * place in d_bytes; no K4 scans, no K5 pass, no size read-back before the blob is written.  Two kernels share the work:
 *   K3F  levels 3-8, 16-byte aligned rows, stream_size a multiple of 4096 -- or a multiple of 4 and at least 8192, in which
 *        case every stream's short last frame takes a detour through a slot: the bitstream is written straight from the
 *        wave's registers, there are no slots;
 *   K3G  everything else (levels 0-2 = 1152-sample blocks, streams shorter than two frames, any length, unaligned rows,
 *        and -- fa_encode_i64_device -- the two-channel frames of int64 arrays): K3's frame body packs the frame into
 *        the wave's own slot (a few thousand slots in all, cache resident) and the wave moves it once its offset is known.
 * fa_encode_single_pass_supported is 1 for every valid geometry (0 only under FLACARRAY_HIP_SLOTS, the diagnostic switch
 * that sends everything through begin + finish).  The caller provides d_bytes with fa_encode_capacity_bytes() bytes
 * (worst case: every frame VERBATIM; a smaller buffer is accepted -- if the blob does not fit, nothing outside the
 * buffer is written and the call returns FA_ERROR_ALLOC) and a workspace of fa_encode_single_pass_workspace_bytes(); the
 * encoded triple is d_bytes[0, *h_total_bytes), d_starts, d_nbytes.  Same bytes as begin + finish in every case.
 * The call waits on `stream` once (for the error word and the total, which is valid on return); d_bytes / d_starts / d_nbytes
 * are complete in stream order: K3F's header and index kernel may still be queued when the call returns. */
int fa_encode_single_pass_supported(int64_t n_stream, int64_t stream_size, uint32_t level);
int64_t fa_encode_capacity_bytes(int64_t n_stream, int64_t stream_size, uint32_t level);
int64_t fa_encode_single_pass_workspace_bytes(int64_t n_stream, int64_t stream_size, uint32_t level);
int fa_encode_i32_device(const int32_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                         int64_t workspace_bytes, unsigned char* d_bytes, int64_t capacity_bytes, int64_t* d_starts,
                         int64_t* d_nbytes, int64_t* h_total_bytes, int32_t* d_info, void* stream);

/* float32 input, quantisation fused into the single-pass encoder (float32_to_int32 of utils.c:160-243 without the
 * int32 round trip through HBM): a range pre-pass writes d_offsets / d_gains[n_stream] (d_quanta may be NULL = per-stream
 * quanta from the data range), then the encoder quantises every sample where it loads it.  Same bytes, offsets and
 * gains as fa_float32_to_int32_device followed by fa_encode_i32_device.  Only for geometries the single-pass kernel
 * covers (fa_encode_single_pass_supported); FA_ERROR_ENCODE_INIT otherwise, FA_ERROR_NAN_INPUT for a NaN. */
int fa_encode_f32_device(const float* d_data, int64_t n_stream, int64_t stream_size, uint32_t level, const float* d_quanta,
                         void* d_workspace, int64_t workspace_bytes, unsigned char* d_bytes, int64_t capacity_bytes,
                         int64_t* d_starts, int64_t* d_nbytes, float* d_offsets, float* d_gains, int64_t* h_total_bytes,
                         int32_t* d_info, void* stream);

/* Host-pointer form of the fused float32 path: what array_compress does with float32 input (compress.py:50-84:
 * float_to_int, then encode_flac) in ONE pass over PCIe -- the float32 samples go up once, are quantised where the
 * encoder loads them (fa_encode_f32_device; geometries the single-pass kernel does not cover are quantised on the device
 * first), and only the compressed bytes come back.  quanta may be NULL (per-stream quanta from the data range) or
 * hold n_stream values; offsets / gains [n_stream] are outputs; *bytes is malloc()'d as in encode_i32.
 * FA_ERROR_NAN_INPUT for a NaN. */
int fa_encode_f32_host(const float* data, int64_t n_stream, int64_t stream_size, uint32_t level, const float* quanta,
                       int64_t* n_bytes, int64_t* starts, unsigned char** bytes, float* offsets, float* gains);

/* The float64 twin: what array_compress does with float64 input -- float64_to_int64 (utils.c:245-327) followed by
 * encode_i64 (compress.py:50-84) -- in one trip: the float64 samples go up once, are quantised on the device and encoded
 * from there as two-channel streams; the int64 image never crosses PCIe (the two-call form moves 8 B per sample down and
 * up again).  quanta may be NULL; offsets / gains [n_stream] are outputs.  FA_ERROR_NAN_INPUT for a NaN. */
int fa_encode_f64_host(const double* data, int64_t n_stream, int64_t stream_size, uint32_t level, const double* quanta,
                       int64_t* n_bytes, int64_t* starts, unsigned char** bytes, double* offsets, double* gains);

/* Decode with the int -> float restore fused into the decoder's store: what array_decompress_slice does for float data
 * (decode_flac, then int_to_float: decompress.py:107-136, utils.c:329-368) in one trip -- compressed bytes up, floats
 * down; the integers never cross PCIe.  offsets / gains [n_stream] as float_to_int returned them; data is
 * float32 / float64 [n_stream][n_decode]; sample range and error codes as decode_i32 / decode_i64 (frame CRC-16s checked). */
int fa_decode_f32_host(const unsigned char* bytes, const int64_t* starts, const int64_t* nbytes, int64_t n_stream, int64_t stream_size,
                       int64_t first_sample, int64_t last_sample, const float* offsets, const float* gains, float* data);
int fa_decode_f64_host(const unsigned char* bytes, const int64_t* starts, const int64_t* nbytes, int64_t n_stream, int64_t stream_size,
                       int64_t first_sample, int64_t last_sample, const double* offsets, const double* gains, double* data);

/* The same three calls for int64 input (two-channel streams); d_info, if given, holds one
 * FrameInfo per SUBFRAME: [ (stream * frames + frame) * 2 + channel ]. */
int64_t fa_encode_workspace_bytes_i64(int64_t n_stream, int64_t stream_size, uint32_t level);
int fa_encode_i64_device_begin(const int64_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level,
                               void* d_workspace, int64_t workspace_bytes, int64_t* d_starts, int64_t* d_nbytes,
                               int64_t* h_total_bytes, int32_t* d_info, void* stream);
int fa_encode_i64_device_finish(int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                                const int64_t* d_starts, unsigned char* d_bytes, void* stream);
/* ... and the single-pass form (K3G; replaces FLAC__stream_encoder_process_interleaved on two channels,
 * compress.c:482-540) with its capacity and workspace queries. */
int64_t fa_encode_capacity_bytes_i64(int64_t n_stream, int64_t stream_size, uint32_t level);
int64_t fa_encode_single_pass_workspace_bytes_i64(int64_t n_stream, int64_t stream_size, uint32_t level);
int fa_encode_i64_device(const int64_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                         int64_t workspace_bytes, unsigned char* d_bytes, int64_t capacity_bytes, int64_t* d_starts,
                         int64_t* d_nbytes, int64_t* h_total_bytes, int32_t* d_info, void* stream);

/* Decode [first_sample,last_sample) (or everything when either is negative) of n_stream
 * streams.  Exactly one of d_out_i32 / d_out_f32 is non-NULL; with d_out_f32 the int32 ->
 * float32 restore (utils.c:350-368) is fused into the store and d_offsets/d_gains[n_stream]
 * are required.  verify: see fa_set_decode_verify.  Returns the OR of the error bits (synchronises the stream). */
int fa_decode_i32_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                         const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t first_sample,
                         int64_t last_sample, int32_t* d_out_i32, float* d_out_f32, const float* d_offsets,
                         const float* d_gains, void* stream, int verify);

/* The same for two-channel (int64 / float64) streams; with d_out_f64 the int64 -> float64 restore
 * (int64_to_float64, utils.c:329-348) is fused into the final pass. */
int fa_decode_i64_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                         const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t first_sample,
                         int64_t last_sample, int64_t* d_out_i64, double* d_out_f64, const double* d_offsets,
                         const double* d_gains, void* stream, int verify);

/* A decode index: the stream metadata and the byte offset of every frame of one store, computed once (K6) and kept in
 * device memory, for many reads of the same store -- the reference's usage pattern is one decode call per key
 * (array.py:409-449), and without an index every call parses every stream header again.  The store (d_bytes, 16-byte
 * aligned) must stay alive and unchanged while the index exists; channels = 1 (int32 / float32) or 2 (int64 / float64).
 * fa_decode_indexed: n_slices < 0 decodes [first_sample, last_sample) (or everything) of ALL streams, as
 * fa_decode_i32_device does; n_slices >= 0 is the batched random access of fa_decode_slices_i32_device.  d_out_int /
 * d_out_float are int32 / float32 for one channel, int64 / float64 for two (offsets / gains likewise). */
int fa_decode_index_create(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts, const int64_t* d_nbytes,
                           int64_t n_stream, int64_t stream_size, int channels, void** index, void* stream);
void fa_decode_index_destroy(void* index);
int fa_decode_indexed(void* index, int64_t first_sample, int64_t last_sample, int64_t n_slices, const int64_t* slice_stream,
                      const int64_t* slice_first, const int64_t* slice_count, const int64_t* out_offset, void* d_out_int,
                      void* d_out_float, const void* d_offsets, const void* d_gains, void* stream, int verify);

/* fa_decode_indexed for scattered slices whose samples are wanted on the HOST: the first out_bytes bytes of the result
 * are in h_out when the call returns.  For launches the latency decoder serves (up to 4096 frames) the kernel
 * writes them straight into host memory -- into h_out itself when that is pinned and device-visible (fa_pinned_alloc
 * below, or any hipHostMalloc), else into the library's own pinned landing buffer for results of up to 512 KB, copied
 * from there -- and the call synchronises once; otherwise the result is decoded into the caller's device buffer
 * (d_out_int / d_out_float as above, always required) and copied.  What the reference does for one small read:
 * seek_absolute + process_single into the caller's array, decompress.c:281-298. */
int fa_decode_indexed_host(void* index, int64_t n_slices, const int64_t* slice_stream, const int64_t* slice_first,
                           const int64_t* slice_count, const int64_t* out_offset, void* d_out_int, void* d_out_float,
                           const void* d_offsets, const void* d_gains, void* h_out, int64_t out_bytes, void* stream, int verify);

/* Pinned, device-visible host memory for results (hipHostMalloc / hipHostFree behind a C signature); NULL on failure.
 * flacarray_amd keeps a small pool of such blocks behind the numpy arrays its read paths return. */
void* fa_pinned_alloc(int64_t bytes);
void fa_pinned_free(void* p);

/* Integrity check of the decoder.  Every device decode entry point takes `verify`: 1 = re-compute the CRC-16 of each
 * frame the call read and report a mismatch as FA_ERROR_DECODE_PROCESS -- what libFLAC reports through the error
 * callback the reference prints (decompress.c:104-121) --, 0 = do not, negative = the process-wide default set here
 * (returns the previous setting; initially off).  The host-pointer decode_i32 / decode_i64 always check (libFLAC does;
 * FLACARRAY_HIP_HOST_VERIFY=0 turns that off).  The header CRC-8 of every frame is checked in all cases. */
int fa_set_decode_verify(int on);

/* Batched random access: slice i is samples [first[i], first[i]+count[i]) of stream
 * slice_stream[i]; its samples are written at element offset out_offset[i] of the output.
 * The four slice arrays are HOST arrays of length n_slices.  The reference needs one
 * decode_i32 call per slice for this (decompress.py:42-48: one range for all streams). */
int fa_decode_slices_i32_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                                const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t n_slices,
                                const int64_t* slice_stream, const int64_t* slice_first, const int64_t* slice_count,
                                const int64_t* out_offset, int32_t* d_out_i32, float* d_out_f32,
                                const float* d_offsets, const float* d_gains, void* stream, int verify);

int fa_decode_slices_i64_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                                const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t n_slices,
                                const int64_t* slice_stream, const int64_t* slice_first, const int64_t* slice_count,
                                const int64_t* out_offset, int64_t* d_out_i64, double* d_out_f64,
                                const double* d_offsets, const double* d_gains, void* stream, int verify);

/* float32 -> int32 quantisation on device; d_quanta may be NULL.  Returns FA_ERROR_NAN_INPUT
 * if any input is NaN (outputs are then unspecified). */
int fa_float32_to_int32_device(const float* d_input, int64_t n_stream, int64_t stream_size, const float* d_quanta,
                               int32_t* d_output, float* d_offsets, float* d_gains, void* stream);

int fa_int32_to_float32_device(const int32_t* d_input, int64_t n_stream, int64_t stream_size, const float* d_offsets,
                               const float* d_gains, float* d_output, void* stream);

/* The float64 <-> int64 twins (utils.c:245-348) on device pointers. */
int fa_float64_to_int64_device(const double* d_input, int64_t n_stream, int64_t stream_size, const double* d_quanta,
                               int64_t* d_output, double* d_offsets, double* d_gains, void* stream);
int fa_int64_to_float64_device(const int64_t* d_input, int64_t n_stream, int64_t stream_size, const double* d_offsets,
                               const double* d_gains, double* d_output, void* stream);

/* Kernel timing for bench.py: when enabled, HIP events are recorded on the launch stream around
 * the three dominant kernels of the most recent calls; fa_profile_last waits for them and
 * returns milliseconds {encode_frames_kernel, compact_frames_kernel, decode_frames_kernel<8>}
 * (-1 for a kernel that has not run). */
void fa_profile_enable(int on);
int fa_profile_last(float* ms3);
/* The same with more pairs: ms[0..n) = {encode_frames_kernel, compact_frames_kernel, decode_frames_kernel<8>,
 * whole encode sequence (first launch of _begin .. last launch of _finish, host gaps included),
 * whole decode sequence (K6 + K7), float32_to_int32_kernel}; n <= 6. */
int fa_profile_read(float* ms, int n);

/* free the library's cached device scratch (decode tables, staging buffers) */
void fa_release_scratch(void);

/* number of visible HIP devices (0 when there is no GPU); never initialises a context */
int fa_device_count(void);

/* library version string */
const char* fa_version(void);

/* ABI revision of the group-2 (fa_*) signatures in this header.  It is raised whenever an existing entry point changes
 * its argument list (revision 2: the trailing `int verify` of the device decode entry points, 1 / 0 / negative = check /
 * do not / process default, see fa_set_decode_verify); a binding built against
 * another revision must refuse the library instead of calling it with a shifted argument list --
 * flacarray_amd/_lib.py does. */
#define FA_ABI_VERSION 2  /* (new entry points do not raise it: fa_encode_f64_host came with revision 2) */
int fa_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLACARRAY_HIP_H */
