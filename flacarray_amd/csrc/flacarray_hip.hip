// flacarray_hip.hip -- host side of the C ABI declared in include/flacarray_hip.h.
//
// Plumbing semantics follow the reference's C layer: argument validation and error bits of
// encode() (src/flacarray/libflacarray/compress.c:133-156) and decode()
// (decompress.c:194-222), malloc()'d output blob owned by the caller (compress.c:251,414),
// starts = exclusive prefix sum of stream sizes (compress.c:402-429).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include "../../include/flacarray_hip.h"
#include "decode_kernels.hpp"
#ifndef FA_SPLIT_UNITS
#define FA_HAVE_K5_LAUNCHERS 1  // single-unit build: the K5 kernels and their launchers live here
#endif
#include "encode_kernels.hpp"
#include "encode_fused.hpp"
#include "encode_placed.hpp"
#include "decode_latency.hpp"
#include "quantize_kernels.hpp"
#include "verify_kernels.hpp"

namespace {

using namespace fa;

#define FA_HIP_TRY(expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "flacarray_hip: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
            return FA_ERROR_DEVICE;                                                               \
        }                                                                                         \
    } while (0)

// Per-device state.  Every compute entry point runs under the api_mu of the CURRENT device (the calls of one
// device share its cached scratch buffers); calls that target different devices -- one process driving several
// GPUs from several threads -- do not serialise each other.  g_mu guards the map of device states only.
constexpr int kProfPairs = 6;
struct DeviceState {
    std::recursive_mutex api_mu;
    std::map<int, float*> windows;  // blocksize -> device tukey(0.5) table
    uint16_t* crc_tab = nullptr;
    uint16_t* crc_tab_fused = nullptr;
    void* scratch[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [10]: K1a partial ranges
    size_t scratch_bytes[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t scratch_epoch = 1;  // bumped whenever a scratch slot is (re)allocated or released: cached contents are then stale
    // frame-header table of the most recent encode geometry (host copy + what the device copy was built from)
    std::vector<uint4> h_hdr;
    int64_t c_nf = -1;
    int c_B = 0, c_tail = 0, c_nch = 0;
    void* c_dp = nullptr;
    uint64_t c_epoch = 0;
    bool stamps_zeroed = false;
    hipStream_t feed_stream = nullptr;  // the host entry points' upload stream (created once: a new stream costs tens of ms)
    // small reads that want their samples on the host: the latency decoder stores them (and its status word) straight
    // into this pinned, device-visible buffer -- no copy calls, one stream synchronisation (decode_device_impl)
    void* pin = nullptr;
    void* pin_dev = nullptr;
    bool pin_tried = false;
    // the frame CRC-16 check of a large decode runs beside K7 on this stream (run_verify_beside)
    hipStream_t verify_stream = nullptr;
    hipEvent_t verify_ev[2] = {nullptr, nullptr};
    bool verify_tried = false;
    // optional in-library kernel timing (HIP events on the launch stream), see fa_profile_enable
    // pairs: 0 K3 encode_frames, 1 K5 compact_frames, 2 K7 decode_frames, 3 whole encode sequence (begin .. finish),
    // 4 whole decode sequence (K6 + K7 + checks), 5 K1 float32_to_int32
    hipEvent_t ev[2 * kProfPairs];
    bool ev_ready = false;
    bool ev_set[kProfPairs] = {false, false, false, false, false, false};
};
std::mutex g_mu;
std::map<int, std::unique_ptr<DeviceState>> g_dev;
std::atomic<bool> g_prof{false};
std::atomic<bool> g_verify{false};  // fa_set_decode_verify: re-compute every decoded frame's CRC-16

DeviceState* dev_state() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    auto& p = g_dev[d];
    if (!p) p.reset(new DeviceState());
    return p.get();
}
#define FA_API_LOCK_OR(fail_stmt)      \
    DeviceState* ds_ = dev_state();    \
    if (!ds_) { fail_stmt; }           \
    std::lock_guard<std::recursive_mutex> api_lock_(ds_->api_mu)
#define FA_API_LOCK FA_API_LOCK_OR(return FA_ERROR_DEVICE)

// the reference's restore functions return void (flacarray.h:295-311): a device failure cannot be reported to the
// caller, and returning garbage silently is worse than stopping -- say which call failed, then abort
[[noreturn]] void fatal_device(const char* fn, const char* what) {
    std::fprintf(stderr, "flacarray_hip: %s: %s failed (%s); this entry point has no error channel and no CPU fallback\n", fn, what,
                 hipGetErrorString(hipGetLastError()));
    std::abort();
}

void prof_begin(int k, hipStream_t st) {
    if (!g_prof) return;
    DeviceState* ds = dev_state();
    if (!ds) return;
    if (!ds->ev_ready) {
        for (auto& e : ds->ev) (void)hipEventCreate(&e);
        ds->ev_ready = true;
    }
    (void)hipEventRecord(ds->ev[2 * k], st);
}
void prof_end(int k, hipStream_t st) {
    if (!g_prof) return;
    DeviceState* ds = dev_state();
    if (!ds) return;
    (void)hipEventRecord(ds->ev[2 * k + 1], st);
    ds->ev_set[k] = true;
}

// grow-only cached device scratch of the current device; slot selects independent buffers (callers hold api_mu)
int get_scratch(int slot, size_t bytes, void** out) {
    DeviceState* st = dev_state();
    if (!st) return FA_ERROR_DEVICE;
    if (st->scratch_bytes[slot] < bytes) {
        if (st->scratch[slot]) (void)hipFree(st->scratch[slot]);
        st->scratch[slot] = nullptr;
        st->scratch_bytes[slot] = 0;
        st->scratch_epoch++;
        size_t want = bytes + (bytes >> 3) + 256;
        if (hipMalloc(&st->scratch[slot], want) != hipSuccess) {
            if (hipMalloc(&st->scratch[slot], bytes) != hipSuccess) return FA_ERROR_ALLOC | FA_ERROR_DEVICE;
            want = bytes;
        }
        st->scratch_bytes[slot] = want;
    }
    *out = st->scratch[slot];
    return FA_ERROR_NONE;
}

// tukey(0.5) window of length L (libFLAC's default apodization for levels 3-5)
void tukey_window(int L, std::vector<float>& w) {
    w.assign((size_t)L, 1.0f);
    const int Np = (int)(0.25f * (float)L) - 1;
    if (Np > 0) {
        for (int n = 0; n <= Np; ++n) {
            w[(size_t)n] = (float)(0.5 - 0.5 * std::cos(3.14159265358979323846 * (double)n / (double)Np));
            w[(size_t)(L - Np - 1 + n)] = (float)(0.5 - 0.5 * std::cos(3.14159265358979323846 * (double)(n + Np) / (double)Np));
        }
    }
}

int get_window(int L, const float** out) {
    DeviceState* st = dev_state();
    if (!st) return FA_ERROR_DEVICE;
    auto it = st->windows.find(L);
    if (it == st->windows.end()) {
        std::vector<float> w;
        tukey_window(L, w);
        float* d = nullptr;
        FA_HIP_TRY(hipMalloc(&d, sizeof(float) * (size_t)(L + 16)));
        FA_HIP_TRY(hipMemset(d, 0, sizeof(float) * (size_t)(L + 16)));
        FA_HIP_TRY(hipMemcpy(d, w.data(), sizeof(float) * (size_t)L, hipMemcpyHostToDevice));
        it = st->windows.emplace(L, d).first;
    }
    *out = it->second;
    return FA_ERROR_NONE;
}

// CRC-16 (poly 0x8005) tables for compact_frames_kernel, see encode_kernels.hpp
int get_crc_tab(const uint16_t** out) {
    DeviceState* st = dev_state();
    if (!st) return FA_ERROR_DEVICE;
    if (!st->crc_tab) {
        std::vector<uint16_t> t((size_t)kCrcTabWords);
        auto feed = [](uint16_t c, uint8_t v) { return crc16_byte(c, v); };
        for (int k = 0; k < 4; ++k)
            for (int v = 0; v < 256; ++v) {
                uint16_t c = 0;
                for (int b = 0; b < 4; ++b) c = feed(c, (uint8_t)(b == k ? v : 0));
                t[(size_t)(k * 256 + v)] = c;
            }
        for (int v = 0; v < 256; ++v) {
            uint16_t hi = (uint16_t)(v << 8), lo = (uint16_t)v;
            for (int b = 0; b < 256; ++b) { hi = feed(hi, 0); lo = feed(lo, 0); }
            t[(size_t)(1024 + v)] = hi;
            t[(size_t)(1280 + v)] = lo;
        }
        uint16_t xp = 1;  // x^0
        for (int n = 0; n < 512; ++n) {
            t[(size_t)(1536 + n)] = xp;
            xp = feed(xp, 0);  // multiply by x^8
        }
        uint16_t* d = nullptr;
        FA_HIP_TRY(hipMalloc(&d, sizeof(uint16_t) * (size_t)kCrcTabWords));
        FA_HIP_TRY(hipMemcpy(d, t.data(), sizeof(uint16_t) * (size_t)kCrcTabWords, hipMemcpyHostToDevice));
        st->crc_tab = d;
    }
    *out = st->crc_tab;
    return FA_ERROR_NONE;
}

// CRC-16 tables of the single-pass encoder (encode_fused.hpp): four slicing tables pre-multiplied by x^2016, so that
// XORing a lane's running state into the top half of its next word (256 bytes further on) advances it for free,
// followed by xpow[i] = x^(8 (i - 255)) mod P for the final per-lane alignment (negative powers through the order of
// x^8 in GF(2)[x] / P, found by iteration).
int get_crc_tab_fused(const uint16_t** out) {
    DeviceState* st = dev_state();
    if (!st) return FA_ERROR_DEVICE;
    if (!st->crc_tab_fused) {
        std::vector<uint16_t> t((size_t)(kFCrcSlice + kFCrcXpow));
        auto mulx8 = [](uint16_t c) { return crc16_byte(c, 0); };
        auto mulmod = [](uint16_t a, uint16_t b) {
            uint32_t r = 0;
            for (int i = 15; i >= 0; --i) {
                r = (r << 1) ^ ((r & 0x8000u) ? 0x18005u : 0u);
                if ((b >> i) & 1) r ^= a;
            }
            return (uint16_t)r;
        };
        int ord = 0;
        for (uint16_t c = 1;;) { c = mulx8(c); ++ord; if (c == 1) break; }
        auto xpow8 = [&](long k) {  // x^(8k) mod P for any integer k
            k %= ord;
            if (k < 0) k += ord;
            uint16_t c = 1;
            for (long i = 0; i < k; ++i) c = mulx8(c);
            return c;
        };
        const uint16_t x2016 = xpow8(252);
        for (int k = 0; k < 4; ++k)
            for (int v = 0; v < 256; ++v) {
                uint16_t c = 0;
                for (int b = 0; b < 4; ++b) c = crc16_byte(c, (uint8_t)(b == k ? v : 0));
                t[(size_t)(k * 256 + v)] = mulmod(c, x2016);
            }
        {
            uint16_t c = xpow8(-255);
            for (int i = 0; i < kFCrcXpow; ++i) { t[(size_t)(kFCrcSlice + i)] = c; c = mulx8(c); }
        }
        uint16_t* d = nullptr;
        FA_HIP_TRY(hipMalloc(&d, sizeof(uint16_t) * t.size()));
        FA_HIP_TRY(hipMemcpy(d, t.data(), sizeof(uint16_t) * t.size(), hipMemcpyHostToDevice));
        st->crc_tab_fused = d;
    }
    *out = st->crc_tab_fused;
    return FA_ERROR_NONE;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct EncodePlan {
    LevelParams P;
    int64_t nf, F;
    int tail_bs;
    size_t off_slots, off_fbytes, off_foff, off_snb, off_total, total;
    int64_t slot_stride;
};

int make_plan(int64_t n_stream, int64_t stream_size, uint32_t level, EncodePlan* pl, int nch = 1) {
    if (level > 8) return FA_ERROR_INVALID_LEVEL;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_ZERO_STREAMSIZE;
    pl->P = level_params(level);
    const int64_t B = pl->P.blocksize;
    pl->nf = (stream_size + B - 1) / B;
    pl->tail_bs = (int)(stream_size - (pl->nf - 1) * B);
    if (18 * pl->nf >= (1 << 24)) return FA_ERROR_ENCODE_PROCESS;  // SEEKTABLE block length is 24 bit
    if (pl->nf > 0x7fffffffLL / n_stream) return FA_ERROR_ENCODE_PROCESS;  // grid limit; host API chunks
    pl->F = n_stream * pl->nf;
    size_t o = 0;
    pl->slot_stride = (int64_t)kSlotBytes * nch;
    pl->off_slots = o;  o = align_up(o + (size_t)pl->F * (size_t)pl->slot_stride, 256);
    pl->off_fbytes = o; o = align_up(o + (size_t)pl->F * 4, 256);
    pl->off_foff = o;   o = align_up(o + (size_t)pl->F * 8, 256);
    pl->off_snb = o;    o = align_up(o + (size_t)n_stream * 8, 256);
    pl->off_total = o;  o = align_up(o + 8, 256);
    pl->total = o + 4096;  // slack: the compaction kernel reads whole groups of 256-byte blocks past a frame's end
    return FA_ERROR_NONE;
}

template <int MLO, int NCH>
void launch_encode(const EncodeArgs& a, int64_t F, hipStream_t st) {
    hipLaunchKernelGGL((encode_frames_kernel<MLO, NCH>), dim3((unsigned)F), dim3(64), 0, st, a);
}

// ---- single-pass encode (encode_fused.hpp): every frame is a full 4096-sample mono frame ----
struct FusedPlan {
    LevelParams P;
    int64_t nf, F, hb;
    size_t off_fbytes, off_fabs, off_zero, off_size, off_off, off_ticket, zero_bytes, off_total, total;
    size_t off_tslots, off_tbytes, off_toff, off_tzero;  // short last frames: slots and the compaction's arguments
    int tail_bs;
    int64_t capacity;
};

static bool slots_forced() { return std::getenv("FLACARRAY_HIP_SLOTS") != nullptr; }  // diagnostic: K3 + K4 + K5 for everything
// f32: float32 input (quantised in the staging load of K3F only: whole frames).  int32 streams may end in a short
// frame -- the slot encoder writes those, K3F the rest -- if every frame still starts on a 16-byte boundary.
bool fused_geometry(int64_t n_stream, int64_t stream_size, uint32_t level, bool f32 = false) {
    if (level < 3 || level > 8 || n_stream <= 0 || stream_size <= 0) return false;
    if (stream_size % kMaxBlock != 0 && (f32 || stream_size % 4 != 0 || stream_size < 2 * kMaxBlock)) return false;
    const int64_t nf = (stream_size + kMaxBlock - 1) / kMaxBlock;
    if (18 * nf >= (1 << 24)) return false;
    if (nf > 0x7fffffffLL / n_stream) return false;
    return !slots_forced();
}

void make_fused_plan(int64_t n_stream, int64_t stream_size, uint32_t level, FusedPlan* pl) {
    pl->P = level_params(level);
    pl->nf = (stream_size + kMaxBlock - 1) / kMaxBlock;
    pl->tail_bs = (int)(stream_size - (pl->nf - 1) * (int64_t)kMaxBlock);
    pl->F = n_stream * pl->nf;
    pl->hb = stream_header_bytes(pl->nf);
    size_t o = 0;
    pl->off_fbytes = o; o = align_up(o + (size_t)pl->F * 4, 256);
    pl->off_fabs = o;   o = align_up(o + (size_t)pl->F * 8, 256);
    pl->off_zero = o;   // everything from here to off_total is zeroed before every launch
    pl->off_size = o;   o = align_up(o + (size_t)pl->F * 4, 256);
    pl->off_off = o;    o = align_up(o + (size_t)pl->F * 8, 256);
    pl->off_ticket = o; o = align_up(o + 16, 256);  // ticket word, error flags
    pl->zero_bytes = o - pl->off_zero;
    pl->off_total = o;  o = align_up(o + 8, 256);
    pl->off_tslots = pl->off_tbytes = pl->off_toff = pl->off_tzero = o;
    if (pl->tail_bs != kMaxBlock) {
        pl->off_tslots = o; o = align_up(o + (size_t)n_stream * (size_t)kSlotBytes + 4096, 256);  // (+ the compaction's group reads)
        pl->off_tbytes = o; o = align_up(o + (size_t)n_stream * 4, 256);
        pl->off_toff = o;   o = align_up(o + (size_t)n_stream * 8, 256);
        pl->off_tzero = o;  o = align_up(o + (size_t)n_stream * 8, 256);
    }
    pl->total = o;
    pl->capacity = pl->F * (int64_t)kSlotBytes + n_stream * pl->hb;
}


// ---- single-pass encode of every other geometry (encode_placed.hpp): K3's frame body, frames placed by their waves ----
struct PlacedPlan {
    LevelParams P;
    int64_t nf, F, hb, slot_stride;
    int tail_bs;
    size_t off_fbytes, off_fabs, off_zero, off_size, off_off, off_ticket, zero_bytes, off_total, off_slots, total;
    int64_t capacity;
};

// the persistent grid (+ the scanner's workgroup), never more than frames + scanner
constexpr int64_t kPlacedBelowFrames = 4096;  // (see placed_preferred)
int64_t placed_grid(int64_t F) {
    int64_t g = kPlacedGrid;
    if (const char* e = std::getenv("FLACARRAY_HIP_PLACED_GRID")) {  // diagnostic: another grid (64 .. 16384 workgroups)
        const long v = std::atol(e);
        if (v >= 64 && v <= 16384) g = v;
    }
    return std::min<int64_t>(g, F) + 1;
}

int make_placed_plan(int64_t n_stream, int64_t stream_size, uint32_t level, int nch, PlacedPlan* pl) {
    if (level > 8) return FA_ERROR_INVALID_LEVEL;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_ZERO_STREAMSIZE;
    pl->P = level_params(level);
    const int64_t B = pl->P.blocksize;
    pl->nf = (stream_size + B - 1) / B;
    pl->tail_bs = (int)(stream_size - (pl->nf - 1) * B);
    if (18 * pl->nf >= (1 << 24)) return FA_ERROR_ENCODE_PROCESS;  // SEEKTABLE block length is 24 bit
    if (pl->nf > 0x7fffffffLL / n_stream) return FA_ERROR_ENCODE_PROCESS;  // 32-bit frame numbers (tickets: + the grid, still 32 bit); the host API chunks
    pl->F = n_stream * pl->nf;
    pl->hb = stream_header_bytes(pl->nf);
    pl->slot_stride = (int64_t)kSlotBytes * nch;
    // (slots sized for the 1152-sample blocks of levels 0-2 -- 5 / 9.5 KB, L2 resident -- change nothing: 11.89 against 11.90 ms)
    size_t o = 0;
    pl->off_fbytes = o; o = align_up(o + (size_t)pl->F * 4, 256);
    pl->off_fabs = o;   o = align_up(o + (size_t)pl->F * 8, 256);
    pl->off_zero = o;   // everything from here to off_total is zeroed before every launch
    pl->off_size = o;   o = align_up(o + (size_t)pl->F * 4, 256);
    pl->off_off = o;    o = align_up(o + (size_t)pl->F * 8, 256);
    pl->off_ticket = o; o = align_up(o + 32, 256);  // ticket word, error flags, the scanner's total
    pl->zero_bytes = o - pl->off_zero;
    pl->off_total = o;  o = align_up(o + 8, 256);
    // two slots per workgroup of the persistent grid + the placement copy's reads past the last slot's end
    pl->off_slots = o;  o = align_up(o + (size_t)placed_grid(pl->F) * 2 * (size_t)pl->slot_stride + 256 * (size_t)(FA_PG_GROUP) + 256, 256);
    pl->total = o;
    pl->capacity = pl->F * (int64_t)kSlotBytes * nch + n_stream * pl->hb;  // (the figure of the slot sequence, whatever the slots here)
    return FA_ERROR_NONE;
}

// optional CRC-16 check of every frame the decode just read (verify_kernels.hpp); h_err receives the refreshed flags
// verify: 1 = check, 0 = do not, negative = the process default (fa_set_decode_verify)
int run_verify(const DecodeArgs& a, int* d_err, int* h_err, hipStream_t st, int verify) {
    if (!(verify < 0 ? g_verify.load() : verify != 0)) return FA_ERROR_NONE;
    const uint16_t* tab = nullptr;
    int rc = get_crc_tab_fused(&tab);
    if (rc) return rc;
    hipLaunchKernelGGL(verify_crc16_kernel, dim3((unsigned)((a.n_tasks + 3) / 4)), dim3(256), 0, st, a, tab);
    FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    return FA_ERROR_NONE;
}

// The same check, issued BESIDE K7 instead of after it: K9 reads only the compressed bytes and the frame table, K7 is
// bound by the latency of its Rice chain at two waves per SIMD and leaves half of the HBM bandwidth and a third of the
// issue slots idle, and K9's waves (few registers) fit beside K7's.  begin: the side stream waits for everything queued
// on `st` so far (K6's tables), then K9 is launched on it; end: `st` waits for K9.  Both kernels report through atomics
// on the same status word.  The side stream has the lowest priority and K9 is queued AFTER K7 (queued first, its 262 144
// small workgroups take every CU and K7 starts when they are done: 7.2 + 2.9 ms, measured).  begin returns false when the
// side stream cannot be had: the caller then checks after K7 as before.
bool run_verify_beside_begin(hipStream_t st) {
    DeviceState* ds = dev_state();
    if (!ds) return false;
    if (!ds->verify_tried) {
        ds->verify_tried = true;
        hipStream_t vs = nullptr;
        int least = 0, greatest = 0;  // (numerically: least priority = the larger number)
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const char* pe = std::getenv("FLACARRAY_HIP_VERIFY_PRIO");  // experiment: "high" / "normal" instead of the lowest priority
        int prio = least;
        if (pe && pe[0] == 'h') prio = greatest;
        else if (pe && pe[0] == 'n') prio = (least + greatest) / 2;
        if (hipStreamCreateWithPriority(&vs, hipStreamNonBlocking, prio) == hipSuccess) {
            if (hipEventCreateWithFlags(&ds->verify_ev[0], hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ds->verify_ev[1], hipEventDisableTiming) == hipSuccess) {
                ds->verify_stream = vs;
            } else {
                (void)hipStreamDestroy(vs);
            }
        }
        if (!ds->verify_stream) (void)hipGetLastError();
    }
    if (!ds->verify_stream) return false;
    if (hipEventRecord(ds->verify_ev[0], st) != hipSuccess) return false;
    return hipStreamWaitEvent(ds->verify_stream, ds->verify_ev[0], 0) == hipSuccess;
}
// (called after K7 has been queued on `st`: K7's workgroups take the chip first, K9's fill what they leave)
void run_verify_beside_launch(const DecodeArgs& a, hipStream_t st) {
    DeviceState* ds = dev_state();
    const uint16_t* tab = nullptr;
    if (!ds || !ds->verify_stream || get_crc_tab_fused(&tab)) return;
    hipLaunchKernelGGL(verify_crc16_kernel, dim3((unsigned)((a.n_tasks + 3) / 4)), dim3(256), 0, ds->verify_stream, a, tab);
    (void)hipEventRecord(ds->verify_ev[1], ds->verify_stream);
    (void)hipStreamWaitEvent(st, ds->verify_ev[1], 0);
}

constexpr size_t kPinBytes = 512u << 10;  // samples (larger results go by DMA: 1.6 MB took 307 us this way, 298 by copy); 64 bytes of status words follow
bool pinned_landing(void** host, void** dev) {
    DeviceState* ds = dev_state();
    if (!ds) return false;
    if (!ds->pin_tried) {
        ds->pin_tried = true;
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, kPinBytes + 64, hipHostMallocDefault) == hipSuccess) {
            if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) { ds->pin = h; ds->pin_dev = d; }
            else (void)hipHostFree(h);
        } else {
            (void)hipGetLastError();
        }
    }
    *host = ds->pin;
    *dev = ds->pin_dev;
    return ds->pin != nullptr;
}

// K7L (one wavefront per frame) is used for launches of at most 4096 frames; above it the throughput decoder's 64 frames
// per wave win (K7L holds ~34 KB of LDS per frame: ~1000 frames in flight).  Measured on the benchmark data (one Rice
// partition per frame, 45-95 us each; tools/lat_crossover.py): 4000 slices = ~6000 frames 0.79 ms against K7's 1.18,
// 8000 slices 1.45 against 0.93 -- a crossover near 8000 frames; frames of 32 partitions cost K7L 0.35-0.5 ms each
// (tools/lat_sweep.py) and cross over near 2000.  4096 limits what either kind of data can lose to ~0.5 ms.
// FLACARRAY_HIP_LATENCY=0 disables it, =1 forces it for every launch of up to 65535 frames (tests).
bool latency_allowed(int64_t n_tasks) {
    const char* e = std::getenv("FLACARRAY_HIP_LATENCY");  // (read per call: the tests switch it)
    if (e && e[0] == '0') return false;
    if (e && e[0] == '1') return n_tasks <= 65535;
    return n_tasks <= 4096;
}

// A decode index: what K6 derives from a store (stream metadata, the byte offset of every frame), kept in device memory
// of its own so that many reads of one store -- the reference's usage pattern, array.py:409-449 -- do not re-parse
// 4096 stream headers and rebuild a million-entry frame table per call (fa_decode_index_create).
struct DecodeIndex {
    const unsigned char* bytes = nullptr;  // the store (owned by the caller, must outlive the index)
    int64_t n_bytes = 0, n_stream = 0, stream_size = 0, nf = 0;
    int32_t B = 0, nch = 1;
    StreamMeta* meta = nullptr;
    int64_t* ftab = nullptr;
    int* err = nullptr;       // 64 ints
    void* tasks = nullptr;    // task table of the scattered-slice calls (grown on demand)
    size_t tasks_bytes = 0;
    int device = -1;
};

// idx == nullptr: parse + index into the cached scratch, then decode (one-off calls).
// idx != nullptr, build_only: parse + index into buffers owned by *idx, no decode.
// idx != nullptr, !build_only: decode with the index (d_bytes / d_starts / d_nbytes are not looked at).
int decode_device_impl(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts, const int64_t* d_nbytes,
                       int64_t n_stream, int64_t stream_size, int64_t first_decode, int64_t n_decode, int64_t n_slices,
                       const int64_t* slice_stream, const int64_t* slice_first, const int64_t* slice_count,
                       const int64_t* out_offset, int32_t* d_out_i32, float* d_out_f32, const float* d_offsets,
                       const float* d_gains, hipStream_t st, int nch = 1, int64_t* d_out_i64 = nullptr, double* d_out_f64 = nullptr,
                       const double* d_offsets64 = nullptr, const double* d_gains64 = nullptr, DecodeIndex* idx = nullptr,
                       bool build_only = false, int verify = -1, void* h_copy = nullptr, size_t h_copy_bytes = 0,
                       bool* h_copied = nullptr) {
    int rc = FA_ERROR_NONE;
    int h_err[4] = {0, 0, 0, 0};
    bool err_cleared = true;  // (the one-off path clears them before K6)
    StreamMeta* d_meta = nullptr;
    int64_t* d_ftab = nullptr;
    int* d_err = nullptr;
    int32_t B = 0;
    int64_t nf = 0;
    const bool use_index = (idx != nullptr) && !build_only;
    if (use_index) {
        d_bytes = idx->bytes; n_bytes = idx->n_bytes; n_stream = idx->n_stream; stream_size = idx->stream_size;
        d_meta = idx->meta; d_ftab = idx->ftab; d_err = idx->err; B = idx->B; nf = idx->nf;
        if (idx->nch != nch) return FA_ERROR_DECODE_INIT;
        err_cleared = false;
        prof_begin(4, st);
        // (the status words are cleared where they are first needed: a small read that lands in pinned host memory
        // never looks at them, and a memset is a launch of its own)
    } else {
    // the decode kernel issues 16-byte loads relative to the blob base: realign if necessary
    if (build_only && (reinterpret_cast<uintptr_t>(d_bytes) & 15)) return FA_ERROR_DECODE_INIT;  // (an index refers to the caller's bytes)
    if (reinterpret_cast<uintptr_t>(d_bytes) & 15) {
        void* al = nullptr;
        int rc0 = get_scratch(7, (size_t)n_bytes + 256, &al);
        if (rc0) return rc0;
        FA_HIP_TRY(hipMemcpyAsync(al, d_bytes, (size_t)n_bytes, hipMemcpyDeviceToDevice, st));
        d_bytes = reinterpret_cast<const unsigned char*>(al);
    }
    // ---- K6: parse stream headers ----
    prof_begin(4, st);
    void* p = nullptr;
    const size_t meta_bytes = align_up((size_t)n_stream * sizeof(StreamMeta), 256);
    if (build_only) {
        FA_HIP_TRY(hipMalloc(&p, meta_bytes + 256 + (size_t)n_stream * 4));
        idx->meta = reinterpret_cast<StreamMeta*>(p);  // (fa_decode_index_destroy frees what is set, also after an error)
    } else {
        rc = get_scratch(1, meta_bytes + 256 + (size_t)n_stream * 4, &p);
        if (rc) return rc;
    }
    d_meta = reinterpret_cast<StreamMeta*>(p);
    // [0]=err [1]=variant flags [2]=streams without a seek table [3]=scan passes of the longest of them
    d_err = reinterpret_cast<int*>(reinterpret_cast<char*>(p) + meta_bytes);
    int* d_sflag = d_err + 64;  // per stream: 1 = sync scan ambiguous, walk serially
    FA_HIP_TRY(hipMemsetAsync(d_err, 0, 32, st));  // ([4]: the latency kernel's "repeat with K7" flag)
    hipLaunchKernelGGL(parse_streams_kernel, dim3((unsigned)((n_stream + 255) / 256)), dim3(256), 0, st, d_bytes, d_starts,
                       d_nbytes, n_stream, stream_size, n_bytes, d_meta, d_err);
    StreamMeta m0;
    FA_HIP_TRY(hipMemcpyAsync(&m0, d_meta, sizeof(StreamMeta), hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    if (h_err[0]) return h_err[0];
    B = m0.B;
    if (m0.channels != nch) return FA_ERROR_DECODE_INIT;  // an int32 stream read as int64 or the reverse
    if (B <= 0 || B > 65535) return FA_ERROR_DECODE_INIT;
    if (B > kMaxBlock * 16) return FA_ERROR_DECODE_INIT;
    nf = (stream_size + B - 1) / B;

    // ---- frame table ----
    void* pt = nullptr;
    if (build_only) {
        FA_HIP_TRY(hipMalloc(&pt, (size_t)n_stream * (size_t)nf * 8 + 256));
        idx->ftab = reinterpret_cast<int64_t*>(pt);
    } else {
        rc = get_scratch(2, (size_t)n_stream * (size_t)nf * 8 + 256, &pt);
        if (rc) return rc;
    }
    d_ftab = reinterpret_cast<int64_t*>(pt);
    const int64_t nt = n_stream * nf;
    hipLaunchKernelGGL(build_frame_table_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, d_bytes, d_meta, n_stream,
                       nf, B, nch, d_ftab, d_err);
    if (h_err[2] > 0) {
        const bool scan = (std::getenv("FLACARRAY_HIP_NO_SYNC_SCAN") == nullptr);  // diagnostic: force the serial walk
        if (scan) {
            FA_HIP_TRY(hipMemsetAsync(d_sflag, 0, (size_t)n_stream * 4, st));
            const unsigned ny = (unsigned)(h_err[3] < 1 ? 1 : (h_err[3] > 65535 ? 65535 : h_err[3]));
            hipLaunchKernelGGL(scan_sync_kernel, dim3((unsigned)n_stream, ny), dim3(256), 0, st, d_bytes, n_bytes, d_meta, nf, B,
                               stream_size, d_ftab, d_sflag);
            hipLaunchKernelGGL(check_scan_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, d_meta, n_stream, nf, B,
                               d_ftab, d_sflag);
        }
        hipLaunchKernelGGL(walk_frames_kernel, dim3((unsigned)((n_stream + 63) / 64)), dim3(64), 0, st, d_bytes, n_bytes, d_meta,
                           n_stream, nf, B, stream_size, d_ftab, scan ? d_sflag : (const int*)nullptr, d_err);
    }
    if (build_only) {
        FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(hipStreamSynchronize(st));
        FA_HIP_TRY(hipGetLastError());
        if (h_err[0]) return h_err[0];
        idx->bytes = d_bytes; idx->n_bytes = n_bytes; idx->n_stream = n_stream; idx->stream_size = stream_size; idx->nf = nf;
        idx->B = B; idx->nch = nch; idx->err = d_err;
        return FA_ERROR_NONE;
    }
    }  // (!use_index)

    // ---- K7 ----
    DecodeArgs a;
    std::memset(&a, 0, sizeof a);
    LatInline inl;
    std::memset(&inl, 0, sizeof inl);
    std::vector<int64_t> h_tasks;
    size_t h_tasks_bytes = 0;
    char* d_tasks = nullptr;
    bool tasks_uploaded = false;
    a.blob = d_bytes; a.blob_bytes = n_bytes; a.meta = d_meta; a.ftab = d_ftab; a.nf = nf; a.B = B;
    a.stream_size = stream_size;
    a.out_i32 = d_out_i32; a.out_f32 = d_out_f32; a.offsets = d_offsets; a.gains = d_gains; a.err = d_err;
    if (n_slices < 0) {
        a.f0 = first_decode / B;
        const int64_t f1 = (first_decode + n_decode - 1) / B;
        a.nfr = f1 - a.f0 + 1;
        a.first = first_decode;
        a.n_decode = n_decode;
        a.n_tasks = n_stream * a.nfr;
    } else {
        // scattered slices: one task per (slice, frame)
        int64_t n_tasks = 0;
        for (int64_t i = 0; i < n_slices; ++i) {
            const int64_t s = slice_stream[i], fst = slice_first[i], cnt = slice_count[i];
            if (s < 0 || s >= n_stream || fst < 0 || cnt <= 0 || fst + cnt > stream_size) return FA_ERROR_DECODE_SAMPLE_RANGE;
            n_tasks += (fst + cnt - 1) / B - fst / B + 1;
        }
        a.n_tasks = n_tasks;
        if (a.n_tasks == 0) return FA_ERROR_NONE;
        if (n_tasks <= 8 && B <= kLatMaxBlock && latency_allowed(n_tasks)) {
            // a handful of frames K7L will take: the task table rides in the kernel arguments (no upload, no
            // synchronisation).  The conditions are those of the K7L branch below -- a stream with larger blocks goes
            // straight to K7, which reads the table from memory.
            int64_t t = 0;
            for (int64_t i = 0; i < n_slices; ++i) {
                const int64_t s = slice_stream[i], fst = slice_first[i], cnt = slice_count[i];
                for (int64_t f = fst / B; f <= (fst + cnt - 1) / B; ++f, ++t) {
                    inl.stream[t] = s; inl.frame[t] = f; inl.first[t] = fst; inl.last[t] = fst + cnt; inl.out_off[t] = out_offset[i];
                }
            }
            inl.n = (int32_t)n_tasks;
        }
        // one staging vector, one copy: [stream | frame | first | last | out offset], each n_tasks long
        const size_t stp = align_up((size_t)n_tasks * 8, 256);
        std::vector<int64_t> h(5 * stp / 8);
        int64_t t = 0;
        for (int64_t i = 0; i < n_slices; ++i) {
            const int64_t s = slice_stream[i], fst = slice_first[i], cnt = slice_count[i];
            for (int64_t f = fst / B; f <= (fst + cnt - 1) / B; ++f, ++t) {
                h[0 * stp / 8 + t] = s; h[1 * stp / 8 + t] = f; h[2 * stp / 8 + t] = fst; h[3 * stp / 8 + t] = fst + cnt;
                h[4 * stp / 8 + t] = out_offset[i];
            }
        }
        void* pl = nullptr;
        if (use_index) {
            if (idx->tasks_bytes < 5 * stp) {
                if (idx->tasks) (void)hipFree(idx->tasks);
                idx->tasks = nullptr; idx->tasks_bytes = 0;
                FA_HIP_TRY(hipMalloc(&idx->tasks, 5 * stp + 4096));
                idx->tasks_bytes = 5 * stp + 4096;
            }
            pl = idx->tasks;
        } else {
            rc = get_scratch(3, 5 * stp, &pl);
            if (rc) return rc;
        }
        char* c = reinterpret_cast<char*>(pl);
        if (inl.n == 0) {
            FA_HIP_TRY(hipMemcpyAsync(c, h.data(), 5 * stp, hipMemcpyHostToDevice, st));
            FA_HIP_TRY(hipStreamSynchronize(st));  // the host vector goes out of scope
        } else {
            h_tasks.swap(h);  // (uploaded only if K7 has to repeat the launch)
            h_tasks_bytes = 5 * stp;
            d_tasks = c;
        }
        a.task_stream = reinterpret_cast<const int64_t*>(c + 0 * stp);
        a.task_frame = reinterpret_cast<const int64_t*>(c + 1 * stp);
        a.task_first = reinterpret_cast<const int64_t*>(c + 2 * stp);
        a.task_last = reinterpret_cast<const int64_t*>(c + 3 * stp);
        a.task_out_off = reinterpret_cast<const int64_t*>(c + 4 * stp);
    }
    if (a.B > kMaxBlock * 16) return FA_ERROR_DECODE_INIT;
    const unsigned nblk = (unsigned)((a.n_tasks + 63) / 64);
    const bool f32 = (d_out_f32 != nullptr);
    const bool f64 = (d_out_f64 != nullptr);
    if (a.B <= kLatMaxBlock && latency_allowed(a.n_tasks)) {
        LatWide wd;
        wd.out_i64 = d_out_i64; wd.out_f64 = d_out_f64; wd.offsets = d_offsets64; wd.gains = d_gains64;
        auto launch_latency = [&](const DecodeArgs& aa, int* flag) {
            const dim3 grid((unsigned)a.n_tasks), block(64);
            if (nch == 2) {
                if (f64) hipLaunchKernelGGL((decode_latency_kernel<true, 2>), grid, block, 0, st, aa, inl, wd, flag);
                else hipLaunchKernelGGL((decode_latency_kernel<false, 2>), grid, block, 0, st, aa, inl, wd, flag);
            } else {
                if (f32) hipLaunchKernelGGL((decode_latency_kernel<true, 1>), grid, block, 0, st, aa, inl, wd, flag);
                else hipLaunchKernelGGL((decode_latency_kernel<false, 1>), grid, block, 0, st, aa, inl, wd, flag);
            }
        };
        // K7L: one wavefront per frame (decode_latency.hpp).  A launch with fewer frames than the chip has lanes is
        // latency bound in K7 (one lane per frame: ~1 ms whatever the count); frames K7L does not take set the flag
        // and the launch is repeated by K7 below.
        const bool verifying = (verify < 0 ? g_verify.load() : verify != 0);
        void *pin_h = nullptr, *pin_d = nullptr;
        // (the caller's own buffer, if it is pinned and device-visible -- fa_pinned_alloc, any hipHostMalloc --, takes
        // the samples directly whatever their size; otherwise the library's landing buffer, for results up to 512 KB)
        void* direct = nullptr;
        if (nch == 1 && h_copy && h_copy_bytes && !verifying) {
            hipPointerAttribute_t at;
            // (up to 2 MB: beyond that the kernel's 256-byte stores over the link are slower than one DMA copy out of
            // the device buffer; at 16 MB the two cost the same, 1.04 ms per 1000-slice read, a quarter of it the Python list of results)
            if (hipPointerGetAttributes(&at, h_copy) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) {
                if (h_copy_bytes <= (2u << 20)) direct = at.devicePointer;
            } else {
                (void)hipGetLastError();
            }
        }
        if (nch == 1 && h_copy && h_copy_bytes && (direct || h_copy_bytes <= kPinBytes) && !verifying && pinned_landing(&pin_h, &pin_d)) {
            // a small read that wants its samples on the host: the kernel stores them and its status word into pinned
            // host memory; what is left for the host is one synchronisation and a memcpy of a few KB
            DecodeArgs ap = a;
            void* const land = direct ? direct : pin_d;
            if (f32) ap.out_f32 = reinterpret_cast<float*>(land); else ap.out_i32 = reinterpret_cast<int32_t*>(land);
            volatile int* status = reinterpret_cast<volatile int*>(reinterpret_cast<char*>(pin_h) + kPinBytes);
            status[0] = 0;
            int* d_status = reinterpret_cast<int*>(reinterpret_cast<char*>(pin_d) + kPinBytes);
            prof_begin(2, st);
            launch_latency(ap, d_status);
            prof_end(2, st);
            prof_end(4, st);
            FA_HIP_TRY(hipStreamSynchronize(st));
            FA_HIP_TRY(hipGetLastError());
            if (status[0] == 0) {
                if (!direct) std::memcpy(h_copy, pin_h, h_copy_bytes);
                if (h_copied) *h_copied = true;
                return FA_ERROR_NONE;
            }
            // (a frame the latency decoder does not take: the whole launch again, the ordinary way)
        }
        if (!err_cleared) { FA_HIP_TRY(hipMemsetAsync(d_err, 0, 32, st)); err_cleared = true; }
        prof_begin(2, st);
        launch_latency(a, d_err + 4);
        prof_end(2, st);
        prof_end(4, st);
        int h8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (h_copy && h_copy_bytes) {
            // a small read wants its samples on the host: they travel with the status words, one synchronisation for both
            FA_HIP_TRY(hipMemcpyAsync(h_copy, f32 ? (const void*)d_out_f32 : (const void*)d_out_i32, h_copy_bytes, hipMemcpyDeviceToHost, st));
        }
        FA_HIP_TRY(hipMemcpyAsync(h8, d_err, 32, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(hipStreamSynchronize(st));
        FA_HIP_TRY(hipGetLastError());
        if (inl.n > 0 && (h8[4] != 0 || verifying)) {  // K7 and the CRC-16 check read the task table from memory
            FA_HIP_TRY(hipMemcpyAsync(d_tasks, h_tasks.data(), h_tasks_bytes, hipMemcpyHostToDevice, st));
            FA_HIP_TRY(hipStreamSynchronize(st));
            tasks_uploaded = true;
        }
        if (h8[4] != 0 && std::getenv("FLACARRAY_HIP_LATENCY_DEBUG"))
            std::fprintf(stderr, "flacarray_hip: latency decoder gave up (reasons %d) on a launch of %lld frames (first slice: stream %lld, sample %lld); repeating with K7\n",
                         h8[4], (long long)a.n_tasks, (long long)(n_slices > 0 ? slice_stream[0] : -1), (long long)(n_slices > 0 ? slice_first[0] : first_decode));
        if (h8[4] == 0) {
            h_err[0] = h8[0];
            if ((rc = run_verify(a, d_err, h_err, st, verify))) return rc;
            if (h_copied) *h_copied = (h_copy && h_copy_bytes);
            return h_err[0];
        }
    }
    if (!err_cleared) { FA_HIP_TRY(hipMemsetAsync(d_err, 0, 32, st)); err_cleared = true; }
    if (inl.n > 0 && !tasks_uploaded) {  // no K7 launch ever sees a task table that only exists in kernel arguments
        FA_HIP_TRY(hipMemcpyAsync(d_tasks, h_tasks.data(), h_tasks_bytes, hipMemcpyHostToDevice, st));
        FA_HIP_TRY(hipStreamSynchronize(st));
        tasks_uploaded = true;
    }
#ifndef FA_DEV_MINIMAL
    if (nch == 2) {
        // two-channel arrays: task-local planar image (low words), bit 32 of every sample, task status
        a.hib_words = (a.B + 31) / 32;
        const size_t tmp_b = align_up((size_t)a.n_tasks * 2 * (size_t)a.B * 4, 256);
        const size_t hib_b = align_up((size_t)a.n_tasks * 2 * (size_t)a.hib_words * 4, 256);
        const size_t asg_b = align_up((size_t)a.n_tasks * 4, 256);
        void* p8 = nullptr;
        rc = get_scratch(8, tmp_b + hib_b + asg_b, &p8);
        if (rc) return rc;
        a.out_i32 = reinterpret_cast<int32_t*>(p8);
        a.out_f32 = nullptr;
        a.hibits = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(p8) + tmp_b);
        a.assign = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(p8) + tmp_b + hib_b);
        FA_HIP_TRY(hipMemsetAsync(a.assign, 0xFF, (size_t)a.n_tasks * 4, st));
        prof_begin(2, st);
        if (std::getenv("FLACARRAY_HIP_NO_VERBATIM_KERNEL") == nullptr) {  // (diagnostic: K7 alone decodes everything)
            hipLaunchKernelGGL(verbatim_channel0_kernel, dim3((unsigned)a.n_tasks), dim3(256), 0, st, a);
            a.verbatim_done = 1;
        }
        hipLaunchKernelGGL((decode_frames_kernel<8, -1, false, 2>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
        prof_end(2, st);
        FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(hipStreamSynchronize(st));
        if (h_err[1] & kFlagNeed16) hipLaunchKernelGGL((decode_frames_kernel<16, 8, false, 2>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
        if (h_err[1]) {  // a frame left over by the 16-deep pass raises the flag again
            FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
            FA_HIP_TRY(hipStreamSynchronize(st));
        }
        if (h_err[1] & kFlagNeed32) hipLaunchKernelGGL((decode_frames_kernel<32, 16, false, 2>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
        hipLaunchKernelGGL(combine_channels_kernel, dim3((unsigned)a.n_tasks), dim3(256), 0, st, a, d_out_i64, d_out_f64, d_offsets64,
                           d_gains64);
        FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(hipStreamSynchronize(st));
        FA_HIP_TRY(hipGetLastError());
        if ((rc = run_verify(a, d_err, h_err, st, verify))) return rc;
        return h_err[0];
    }
#endif
    prof_begin(2, st);
#ifdef FA_DEV_MINIMAL
    (void)f32;
    hipLaunchKernelGGL((decode_frames_kernel<8, -1, false, 1>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
    prof_end(2, st);
    prof_end(4, st);
    FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    if ((rc = run_verify(a, d_err, h_err, st, verify))) return rc;
    return h_err[0] | (h_err[1] ? FA_ERROR_DECODE_PROCESS : 0);
#else
    // the frame CRC-16 check of a launch that fills the chip goes beside K7 (it needs what K6 built, nothing of K7's)
    const bool verifying_k7 = (verify < 0 ? g_verify.load() : verify != 0);
    const bool beside = verifying_k7 && a.n_tasks >= 16384 && std::getenv("FLACARRAY_HIP_VERIFY_AFTER") == nullptr &&
                        run_verify_beside_begin(st);
    if (f32) hipLaunchKernelGGL((decode_frames_kernel<8, -1, true, 1>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
    else hipLaunchKernelGGL((decode_frames_kernel<8, -1, false, 1>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
    prof_end(2, st);
    if (beside) {
        run_verify_beside_launch(a, st);
        verify = 0;  // (done: the check at the end of this function is not repeated)
    }
    prof_end(4, st);  // (deeper-history passes, when a stream needs them, follow outside this pair)
    FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    if (h_err[1] & kFlagNeed16) {
        if (f32) hipLaunchKernelGGL((decode_frames_kernel<16, 8, true, 1>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
        else hipLaunchKernelGGL((decode_frames_kernel<16, 8, false, 1>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
    }
    if (h_err[1] & kFlagNeed32) {
        if (f32) hipLaunchKernelGGL((decode_frames_kernel<32, 16, true, 1>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
        else hipLaunchKernelGGL((decode_frames_kernel<32, 16, false, 1>), dim3(nblk), dim3(64), 0, st, a, d_err + 1);
    }
    if (h_err[1]) {
        FA_HIP_TRY(hipMemcpyAsync(h_err, d_err, 16, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(hipStreamSynchronize(st));
    }
    FA_HIP_TRY(hipGetLastError());
    if ((rc = run_verify(a, d_err, h_err, st, verify))) return rc;
    return h_err[0];
#endif
}

int validate_range(int64_t stream_size, int64_t first_sample, int64_t last_sample, int64_t* first_decode, int64_t* n_decode) {
    *first_decode = 0;
    *n_decode = stream_size;
    if (first_sample >= 0 && last_sample >= 0) {  // decompress.c:209-222
        if (last_sample > stream_size) return FA_ERROR_DECODE_SAMPLE_RANGE;
        if (first_sample > stream_size - 1) return FA_ERROR_DECODE_SAMPLE_RANGE;
        if (first_sample >= last_sample) return FA_ERROR_DECODE_SAMPLE_RANGE;
        *first_decode = first_sample;
        *n_decode = last_sample - first_sample;
    }
    return FA_ERROR_NONE;
}

}  // namespace

extern "C" {

const char* fa_version(void) { return "flacarray_hip 0.1.0 (gfx950)"; }
int fa_abi_version(void) { return FA_ABI_VERSION; }

int fa_set_decode_verify(int on) {
    const bool was = g_verify.exchange(on != 0);
    return was ? 1 : 0;
}

void fa_profile_enable(int on) { g_prof = (on != 0); }  // process-wide switch; the events are per device

#ifdef FA_STAMPS
// diagnostic build only (-DFA_STAMPS, flacarray_amd/build.py --stamps): read (and optionally clear) the per-phase
// cycle sums of K3.  Not part of the shipped ABI.
int fa_debug_stamps(unsigned long long* out32, int reset) {
    FA_API_LOCK;
    void* sp = nullptr;
    if (get_scratch(6, 512, &sp)) return FA_ERROR_DEVICE;
    if (hipDeviceSynchronize() != hipSuccess) return FA_ERROR_DEVICE;
    if (hipMemcpy(out32, sp, 512, hipMemcpyDeviceToHost) != hipSuccess) return FA_ERROR_DEVICE;
    if (reset && hipMemset(sp, 0, 512) != hipSuccess) return FA_ERROR_DEVICE;
    return FA_ERROR_NONE;
}
#endif

int fa_profile_read(float* ms, int n) {
    FA_API_LOCK;
    for (int k = 0; k < n; ++k) {
        ms[k] = -1.0f;
        if (k < kProfPairs && ds_->ev_ready && ds_->ev_set[k]) {
            if (hipEventSynchronize(ds_->ev[2 * k + 1]) != hipSuccess) return FA_ERROR_DEVICE;
            float t = 0.0f;
            if (hipEventElapsedTime(&t, ds_->ev[2 * k], ds_->ev[2 * k + 1]) == hipSuccess) ms[k] = t;
        }
    }
    return FA_ERROR_NONE;
}

int fa_profile_last(float* ms3) { return fa_profile_read(ms3, 3); }

int fa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void fa_release_scratch(void) {
    FA_API_LOCK_OR(return);
    for (int i = 0; i < 12; ++i) {
        if (ds_->scratch[i]) (void)hipFree(ds_->scratch[i]);
        ds_->scratch[i] = nullptr;
        ds_->scratch_bytes[i] = 0;
    }
    ds_->scratch_epoch++;
    if (ds_->pin) (void)hipHostFree(ds_->pin);
    ds_->pin = ds_->pin_dev = nullptr;
    ds_->pin_tried = false;
}

int64_t fa_encode_workspace_bytes(int64_t n_stream, int64_t stream_size, uint32_t level) {
    EncodePlan pl;
    if (make_plan(n_stream, stream_size, level, &pl) != FA_ERROR_NONE) return -1;
    return (int64_t)pl.total;
}

int64_t fa_encode_workspace_bytes_i64(int64_t n_stream, int64_t stream_size, uint32_t level) {
    EncodePlan pl;
    if (make_plan(n_stream, stream_size, level, &pl, 2) != FA_ERROR_NONE) return -1;
    return (int64_t)pl.total;
}

static int encode_device_begin(const int32_t* d_data, int nch, int64_t n_stream, int64_t stream_size, uint32_t level,
                               void* d_workspace, int64_t workspace_bytes, int64_t* d_starts, int64_t* d_nbytes,
                               int64_t* h_total_bytes, int32_t* d_info, void* stream) {
    FA_API_LOCK;
    EncodePlan pl;
    int rc = make_plan(n_stream, stream_size, level, &pl, nch);
    if (rc) return rc;
    if (!d_workspace || workspace_bytes < (int64_t)pl.total) return FA_ERROR_ALLOC;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(d_workspace);
    prof_begin(3, st);
    EncodeArgs a;
    std::memset(&a, 0, sizeof a);
    a.data = d_data; a.n_stream = n_stream; a.stream_size = stream_size; a.nframes = pl.nf;
    a.B = pl.P.blocksize; a.tail_bs = pl.tail_bs;
    a.max_lpc_order = pl.P.max_lpc_order; a.max_porder = pl.P.max_porder; a.precision = pl.P.qlp_precision;
    rc = get_window(a.B, &a.win);
    if (rc) return rc;
    rc = get_window(a.tail_bs, &a.win_tail);
    if (rc) return rc;
    a.slots = reinterpret_cast<uint8_t*>(ws + pl.off_slots);
    a.slot_stride = pl.slot_stride;
    a.frame_bytes = reinterpret_cast<uint32_t*>(ws + pl.off_fbytes);
    a.info = reinterpret_cast<FrameInfo*>(d_info);
    a.stamps = nullptr;
    a.pmax_full = max_porder_for(a.B, a.max_porder, 0);
    a.pmax_tail = max_porder_for(a.tail_bs, a.max_porder, 0);
    a.escale_full = 0.5 / (double)a.B;
    a.escale_tail = 0.5 / (double)a.tail_bs;
    {
        // frame header fields by frame number: tabulated on the host, cached on the device (per-device state)
        void* dp = nullptr;
        const size_t ntab = (size_t)pl.nf * (nch == 2 ? 2 : 1);  // two-channel arrays: a second half with assignment side + right
        rc = get_scratch(9, ntab * sizeof(uint4) + 256, &dp);
        if (rc) return rc;
        if (ds_->c_nf != pl.nf || ds_->c_B != a.B || ds_->c_tail != a.tail_bs || ds_->c_nch != nch || ds_->c_dp != dp ||
            ds_->c_epoch != ds_->scratch_epoch) {
            ds_->h_hdr.resize(ntab);
            for (int64_t f = 0; f < pl.nf; ++f) {
                ds_->h_hdr[(size_t)f] = frame_header_entry((uint64_t)f, (f == pl.nf - 1) ? a.tail_bs : a.B, nch);
                if (nch == 2) ds_->h_hdr[(size_t)(pl.nf + f)] = frame_header_entry((uint64_t)f, (f == pl.nf - 1) ? a.tail_bs : a.B, nch, true);
            }
            FA_HIP_TRY(hipMemcpyAsync(dp, ds_->h_hdr.data(), ntab * sizeof(uint4), hipMemcpyHostToDevice, st));
            FA_HIP_TRY(hipStreamSynchronize(st));  // h_hdr is reused by the next call
            ds_->c_nf = pl.nf; ds_->c_B = a.B; ds_->c_tail = a.tail_bs; ds_->c_nch = nch; ds_->c_dp = dp; ds_->c_epoch = ds_->scratch_epoch;
        }
        a.hdr = reinterpret_cast<const uint4*>(dp);
    }
#ifdef FA_STAMPS
    {
        void* sp = nullptr;
        if (get_scratch(6, 512, &sp) == 0) {
            if (!ds_->stamps_zeroed) { (void)hipMemset(sp, 0, 512); ds_->stamps_zeroed = true; }
            a.stamps = reinterpret_cast<unsigned long long*>(sp);
        }
    }
#endif
    prof_begin(0, st);
#ifdef FA_DEV_MINIMAL  // diagnostic builds (seconds instead of minutes to compile): level 3-5 int32 kernels only
    launch_encode<8, 1>(a, pl.F, st);
#else
    if (nch == 1) {
        switch (a.max_lpc_order) {
            case 0: launch_encode<0, 1>(a, pl.F, st); break;
            case 6: launch_encode<6, 1>(a, pl.F, st); break;
            case 8: launch_encode<8, 1>(a, pl.F, st); break;
            default: launch_encode<12, 1>(a, pl.F, st); break;
        }
    } else {
        switch (a.max_lpc_order) {
            case 0: launch_encode<0, 2>(a, pl.F, st); break;
            case 6: launch_encode<6, 2>(a, pl.F, st); break;
            case 8: launch_encode<8, 2>(a, pl.F, st); break;
            default: launch_encode<12, 2>(a, pl.F, st); break;
        }
    }
#endif
    prof_end(0, st);
    int64_t* d_foff = reinterpret_cast<int64_t*>(ws + pl.off_foff);
    int64_t* d_snb = reinterpret_cast<int64_t*>(ws + pl.off_snb);
    int64_t* d_total = reinterpret_cast<int64_t*>(ws + pl.off_total);
    hipLaunchKernelGGL(stream_scan_kernel, dim3((unsigned)n_stream), dim3(256), 0, st, a.frame_bytes, pl.nf, d_foff, d_snb);
    hipLaunchKernelGGL(starts_scan_kernel, dim3(1), dim3(1024), 0, st, d_snb, n_stream, d_starts, d_total);
    FA_HIP_TRY(hipMemcpyAsync(d_nbytes, d_snb, (size_t)n_stream * 8, hipMemcpyDeviceToDevice, st));
    FA_HIP_TRY(hipMemcpyAsync(h_total_bytes, d_total, 8, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    FA_HIP_TRY(hipGetLastError());
    return FA_ERROR_NONE;
}

int fa_encode_i32_device_begin(const int32_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level,
                               void* d_workspace, int64_t workspace_bytes, int64_t* d_starts, int64_t* d_nbytes,
                               int64_t* h_total_bytes, int32_t* d_info, void* stream) {
    return encode_device_begin(d_data, 1, n_stream, stream_size, level, d_workspace, workspace_bytes, d_starts, d_nbytes,
                               h_total_bytes, d_info, stream);
}

int fa_encode_i64_device_begin(const int64_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level,
                               void* d_workspace, int64_t workspace_bytes, int64_t* d_starts, int64_t* d_nbytes,
                               int64_t* h_total_bytes, int32_t* d_info, void* stream) {
    return encode_device_begin(reinterpret_cast<const int32_t*>(d_data), 2, n_stream, stream_size, level, d_workspace,
                               workspace_bytes, d_starts, d_nbytes, h_total_bytes, d_info, stream);
}

static int encode_device_finish(int nch, int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                                const int64_t* d_starts, unsigned char* d_bytes, void* stream) {
    FA_API_LOCK;
    EncodePlan pl;
    int rc = make_plan(n_stream, stream_size, level, &pl, nch);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(d_workspace);
    const uint16_t* crc = nullptr;
    rc = get_crc_tab(&crc);
    if (rc) return rc;
    const int64_t* d_foff = reinterpret_cast<const int64_t*>(ws + pl.off_foff);
    launch_write_headers(st, n_stream, d_bytes, d_starts, d_foff, pl.nf, stream_size, (int32_t)pl.P.blocksize, (int32_t)pl.tail_bs,
                         (int32_t)nch);
    int64_t nblk = (pl.F + 3) / 4;
#ifndef FA_K5_MAXBLK
#define FA_K5_MAXBLK 32768
#endif
    if (nblk > FA_K5_MAXBLK) nblk = FA_K5_MAXBLK;
    prof_begin(1, st);
    launch_compact_frames(st, nblk, reinterpret_cast<const uint8_t*>(ws + pl.off_slots),
                          reinterpret_cast<const uint32_t*>(ws + pl.off_fbytes), d_foff, d_starts, pl.nf, pl.F, crc, d_bytes,
                          pl.slot_stride);
    prof_end(1, st);
    prof_end(3, st);
    FA_HIP_TRY(hipGetLastError());
    return FA_ERROR_NONE;
}

int fa_encode_i32_device_finish(int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                                const int64_t* d_starts, unsigned char* d_bytes, void* stream) {
    return encode_device_finish(1, n_stream, stream_size, level, d_workspace, d_starts, d_bytes, stream);
}

int fa_encode_i64_device_finish(int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                                const int64_t* d_starts, unsigned char* d_bytes, void* stream) {
    return encode_device_finish(2, n_stream, stream_size, level, d_workspace, d_starts, d_bytes, stream);
}

// Every valid geometry has a single-pass encoder: K3F (full mono frames of levels 3-8) or K3G (the rest).
int fa_encode_single_pass_supported(int64_t n_stream, int64_t stream_size, uint32_t level) {
    PlacedPlan pl;
    return (make_placed_plan(n_stream, stream_size, level, 1, &pl) == FA_ERROR_NONE && !slots_forced()) ? 1 : 0;
}

static int64_t capacity_bytes_for(int64_t n_stream, int64_t stream_size, uint32_t level, int nch) {
    PlacedPlan pl;
    if (make_placed_plan(n_stream, stream_size, level, nch, &pl) != FA_ERROR_NONE) return -1;
    return pl.capacity;  // every frame VERBATIM: one slot per frame and channel + the stream headers (K3F's figure is the same)
}
int64_t fa_encode_capacity_bytes(int64_t n_stream, int64_t stream_size, uint32_t level) { return capacity_bytes_for(n_stream, stream_size, level, 1); }
int64_t fa_encode_capacity_bytes_i64(int64_t n_stream, int64_t stream_size, uint32_t level) { return capacity_bytes_for(n_stream, stream_size, level, 2); }

// The workspace serves whichever sequence the call takes: K3F's, K3G's (also what K3F's geometries take when the rows
// are not 16-byte aligned), or -- FLACARRAY_HIP_SLOTS -- the slot sequence's.
static int64_t single_pass_workspace_for(int64_t n_stream, int64_t stream_size, uint32_t level, int nch) {
    if (slots_forced()) return nch == 2 ? fa_encode_workspace_bytes_i64(n_stream, stream_size, level) : fa_encode_workspace_bytes(n_stream, stream_size, level);
    PlacedPlan pp;
    if (make_placed_plan(n_stream, stream_size, level, nch, &pp) != FA_ERROR_NONE) return -1;
    int64_t need = (int64_t)pp.total;
    if (nch == 1 && fused_geometry(n_stream, stream_size, level)) {
        FusedPlan pl;
        make_fused_plan(n_stream, stream_size, level, &pl);
        need = std::max<int64_t>(need, (int64_t)pl.total);
    }
    return need;
}
int64_t fa_encode_single_pass_workspace_bytes(int64_t n_stream, int64_t stream_size, uint32_t level) {
    return single_pass_workspace_for(n_stream, stream_size, level, 1);
}
int64_t fa_encode_single_pass_workspace_bytes_i64(int64_t n_stream, int64_t stream_size, uint32_t level) {
    return single_pass_workspace_for(n_stream, stream_size, level, 2);
}

// the single-pass sequence: (float32 input: range pre-pass K1a/K1b,) zero the publish words, K3F, stream headers
static int fused_encode_run(const void* d_data, bool f32, const float* d_quanta, float* d_offsets, float* d_gains, int64_t n_stream,
                            int64_t stream_size, uint32_t level, void* d_workspace, int64_t workspace_bytes, unsigned char* d_bytes,
                            int64_t capacity_bytes, int64_t* d_starts, int64_t* d_nbytes, int64_t* h_total_bytes, int32_t* d_info,
                            void* stream) {
    FA_API_LOCK;
    FusedPlan pl;
    make_fused_plan(n_stream, stream_size, level, &pl);
    if (!d_workspace || workspace_bytes < (int64_t)pl.total) return FA_ERROR_ALLOC;
    // The buffer may be smaller than the worst case (every frame VERBATIM): a frame whose offset lies outside it is not
    // written and the call reports FA_ERROR_ALLOC -- the caller gambles on its data's compressibility and retries with
    // fa_encode_capacity_bytes() if it loses.  It must at least hold the stream headers.
    if (!d_bytes || capacity_bytes < n_stream * pl.hb + 64) return FA_ERROR_ALLOC;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(d_workspace);
    prof_begin(3, st);
    int* d_nanflag = reinterpret_cast<int*>(ws + pl.off_ticket + 12);  // (inside the zeroed region)
    FA_HIP_TRY(hipMemsetAsync(ws + pl.off_zero, 0, pl.zero_bytes, st));
    if (f32) {
        const int64_t cps = (stream_size + kRangeChunk - 1) / kRangeChunk;
        void* pp = nullptr;
        int rcp = get_scratch(10, (size_t)n_stream * (size_t)cps * 8 + 256, &pp);
        if (rcp) return rcp;
        float* pmin = reinterpret_cast<float*>(pp);
        float* pmax = pmin + n_stream * cps;
        prof_begin(5, st);
        hipLaunchKernelGGL(float32_range_kernel, dim3((unsigned)(n_stream * cps)), dim3(256), 0, st, reinterpret_cast<const float*>(d_data),
                           stream_size, cps, pmin, pmax, d_nanflag);
        hipLaunchKernelGGL(float32_params_kernel, dim3((unsigned)((n_stream + 255) / 256)), dim3(256), 0, st, pmin, pmax, n_stream, cps,
                           d_quanta, d_offsets, d_gains);
        prof_end(5, st);
    }
    FusedArgs a;
    std::memset(&a, 0, sizeof a);
    a.data = reinterpret_cast<const int32_t*>(d_data); a.f_offsets = d_offsets; a.f_gains = d_gains; a.n_stream = n_stream; a.stream_size = stream_size; a.nframes = pl.nf; a.total_frames = pl.F;
    a.max_lpc_order = pl.P.max_lpc_order; a.max_porder = pl.P.max_porder; a.precision = pl.P.qlp_precision;
    a.pmax_full = max_porder_for(kMaxBlock, a.max_porder, 0);
    a.escale_full = 0.5 / (double)kMaxBlock;
    a.tail_bs = pl.tail_bs;
    const bool tails = (pl.tail_bs != kMaxBlock);
    int rc = get_window(kMaxBlock, &a.win);
    if (rc) return rc;
    rc = get_crc_tab_fused(&a.crc_tab);
    if (rc) return rc;
    {
        void* dp = nullptr;
        rc = get_scratch(9, (size_t)pl.nf * sizeof(uint4) + 256, &dp);
        if (rc) return rc;
        if (ds_->c_nf != pl.nf || ds_->c_B != kMaxBlock || ds_->c_tail != pl.tail_bs || ds_->c_nch != 1 || ds_->c_dp != dp ||
            ds_->c_epoch != ds_->scratch_epoch) {
            ds_->h_hdr.resize((size_t)pl.nf);
            for (int64_t f = 0; f < pl.nf; ++f)
                ds_->h_hdr[(size_t)f] = frame_header_entry((uint64_t)f, (f == pl.nf - 1) ? pl.tail_bs : kMaxBlock, 1);
            FA_HIP_TRY(hipMemcpyAsync(dp, ds_->h_hdr.data(), (size_t)pl.nf * sizeof(uint4), hipMemcpyHostToDevice, st));
            FA_HIP_TRY(hipStreamSynchronize(st));
            ds_->c_nf = pl.nf; ds_->c_B = kMaxBlock; ds_->c_tail = pl.tail_bs; ds_->c_nch = 1; ds_->c_dp = dp; ds_->c_epoch = ds_->scratch_epoch;
        }
        a.hdr = reinterpret_cast<const uint4*>(dp);
    }
    a.blob = d_bytes; a.capacity = capacity_bytes; a.hb = pl.hb;
    a.frame_bytes = reinterpret_cast<uint32_t*>(ws + pl.off_fbytes);
    a.frame_abs = reinterpret_cast<int64_t*>(ws + pl.off_fabs);
    a.info = reinterpret_cast<FrameInfo*>(d_info);
    a.size_pub = reinterpret_cast<uint32_t*>(ws + pl.off_size);
    a.off_pub = reinterpret_cast<unsigned long long*>(ws + pl.off_off);
    a.total = reinterpret_cast<int64_t*>(ws + pl.off_ticket + 16);  // (next to the error and NaN words: one copy brings all three back)
    a.ticket = reinterpret_cast<uint32_t*>(ws + pl.off_ticket);
    a.err = reinterpret_cast<int*>(ws + pl.off_ticket + 8);
#ifdef FA_STAMPS
    {
        void* sp = nullptr;
        if (get_scratch(6, 512, &sp) == 0) {
            if (!ds_->stamps_zeroed) { (void)hipMemset(sp, 0, 512); ds_->stamps_zeroed = true; }
            a.stamps = reinterpret_cast<unsigned long long*>(sp);
        }
    }
#endif
    if (tails) {
        // the short last frame of every stream: the slot encoder (one workgroup per stream) writes it to a slot and its
        // size goes into size_pub, so that the scanner places it between its neighbours like any other frame
        EncodeArgs t;
        std::memset(&t, 0, sizeof t);
        t.data = a.data; t.n_stream = n_stream; t.stream_size = stream_size; t.nframes = pl.nf; t.B = kMaxBlock; t.tail_bs = pl.tail_bs;
        t.max_lpc_order = a.max_lpc_order; t.max_porder = a.max_porder; t.precision = a.precision;
        t.win = a.win;
        rc = get_window(pl.tail_bs, &t.win_tail);
        if (rc) return rc;
        t.slots = reinterpret_cast<uint8_t*>(ws + pl.off_tslots); t.slot_stride = kSlotBytes;
        t.frame_bytes = a.frame_bytes; t.info = a.info; t.hdr = a.hdr;
        t.pmax_full = a.pmax_full; t.pmax_tail = max_porder_for(pl.tail_bs, a.max_porder, 0);
        t.escale_full = a.escale_full; t.escale_tail = 0.5 / (double)pl.tail_bs;
        t.tail_only = 1;
#ifdef FA_DEV_MINIMAL
        launch_encode<8, 1>(t, n_stream, st);
#else
        switch (t.max_lpc_order) {
            case 6: launch_encode<6, 1>(t, n_stream, st); break;
            case 8: launch_encode<8, 1>(t, n_stream, st); break;
            default: launch_encode<12, 1>(t, n_stream, st); break;
        }
#endif
        launch_fused_tail_publish(st, a.frame_bytes, a.size_pub, n_stream, pl.nf);
    }
    prof_begin(0, st);
    launch_fused_encode(st, a, f32);
    prof_end(0, st);
    struct { int32_t err, nan; int64_t total; } back = {0, 0, 0};
    static_assert(sizeof back == 16, "error word at +8, NaN flag at +12, total at +16 of the ticket block");
    {
        // A frame that was dropped (no offset in time, or an offset outside the buffer) leaves frame_abs / off_pub of
        // itself -- and, after a scanner time-out, of every frame behind it -- unwritten: the kernels below would
        // turn those into addresses.  They run only after the error word has come back clean (one stream
        // synchronisation, ~20 us against a 15 ms kernel).  Nothing behind this point changes the three words.
        FA_HIP_TRY(hipMemcpyAsync(&back, a.err, sizeof back, hipMemcpyDeviceToHost, st));  // (the scanner's total: headers and tails included)
        FA_HIP_TRY(hipStreamSynchronize(st));
        if (back.err == 1 || (back.err == 0 && back.total > capacity_bytes)) return FA_ERROR_ALLOC;  // the blob does not fit the caller's buffer: nothing else is launched
        if (back.err) {
            std::fprintf(stderr, "flacarray_hip: single-pass encode failed (flags %d: 1 = offset outside the buffer, 2 = a frame timed out waiting for its offset, 4 = the scanner timed out)\n", back.err);
            return FA_ERROR_ENCODE_PROCESS;
        }
    }
    if (tails) {
        // every frame has its offset now: move the short frames from their slots (byte-shifted copy + CRC-16, K5)
        uint32_t* tb = reinterpret_cast<uint32_t*>(ws + pl.off_tbytes);
        int64_t* toff = reinterpret_cast<int64_t*>(ws + pl.off_toff);
        int64_t* tzero = reinterpret_cast<int64_t*>(ws + pl.off_tzero);
        launch_fused_tail_prep(st, a.off_pub, a.frame_bytes, n_stream, pl.nf, a.frame_abs, tb, toff, tzero);
        const uint16_t* crc5 = nullptr;
        rc = get_crc_tab(&crc5);
        if (rc) return rc;
        int64_t nblk = (n_stream + 3) / 4;
        if (nblk > 32768) nblk = 32768;
        launch_compact_frames(st, nblk, reinterpret_cast<const uint8_t*>(ws + pl.off_tslots), tb, toff, tzero, 1, n_stream, crc5, d_bytes, kSlotBytes);
    }
    launch_fused_finish(st, d_bytes, a.frame_abs, a.frame_bytes, n_stream, pl.nf, stream_size, (int32_t)kMaxBlock, (int32_t)pl.tail_bs, 1, pl.hb,
                        d_starts, d_nbytes, a.total);
    prof_end(3, st);
    // No second wait: the error word, the NaN flag and the total are in hand, and what is still queued (a short-frame
    // compaction, the header / index kernel: microseconds) completes in stream order like everything else a caller
    // queues behind this call -- one host wait per encode (0.172 -> 0.157 ms at 4096 frames).
    FA_HIP_TRY(hipGetLastError());
    *h_total_bytes = back.total;
    if (back.nan & 1) return FA_ERROR_NAN_INPUT;
    return FA_ERROR_NONE;
}


// Small arrays of K3F's geometries take K3G too: K3F's sequence is four launches and two host waits (five more launches
// when the streams end in a short frame), K3G's is one launch and one wait, which wins until the frames are many enough
// for K3F's faster frame loop to pay (tools/bench_placed.py).  FLACARRAY_HIP_PLACED_BELOW overrides the frame count.
static bool placed_preferred(int64_t n_stream, int64_t stream_size) {
    // (tools/kb_crossover_placed.py, whole frames: 512 frames 0.089 against 0.096 ms, 1024 frames 0.104 against 0.099, 2048
    // frames 0.144 against 0.115; streams that end in a short frame add five launches to K3F's side: 3000 frames 0.166 against 0.22)
    int64_t below = (stream_size % kMaxBlock == 0) ? kPlacedBelowFrames / 4 : kPlacedBelowFrames;
    if (const char* e = std::getenv("FLACARRAY_HIP_PLACED_BELOW")) below = std::atoll(e);
    return n_stream * ((stream_size + kMaxBlock - 1) / kMaxBlock) < below;
}

// the single-pass sequence of every geometry K3F does not take: zero the publish words, K3G, stream headers
static int placed_encode_run(const int32_t* d_data, int nch, int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                             int64_t workspace_bytes, unsigned char* d_bytes, int64_t capacity_bytes, int64_t* d_starts,
                             int64_t* d_nbytes, int64_t* h_total_bytes, int32_t* d_info, void* stream) {
    FA_API_LOCK;
    PlacedPlan pl;
    int rc = make_placed_plan(n_stream, stream_size, level, nch, &pl);
    if (rc) return rc;
    if (!d_workspace || workspace_bytes < (int64_t)pl.total) return FA_ERROR_ALLOC;
    if (!d_bytes || capacity_bytes < n_stream * pl.hb + 64) return FA_ERROR_ALLOC;  // (as for K3F: the buffer may gamble, but holds the headers)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(d_workspace);
    prof_begin(3, st);
    FA_HIP_TRY(hipMemsetAsync(ws + pl.off_zero, 0, pl.zero_bytes, st));
    EncodeArgs a;
    std::memset(&a, 0, sizeof a);
    a.data = d_data; a.n_stream = n_stream; a.stream_size = stream_size; a.nframes = pl.nf;
    a.B = pl.P.blocksize; a.tail_bs = pl.tail_bs;
    a.max_lpc_order = pl.P.max_lpc_order; a.max_porder = pl.P.max_porder; a.precision = pl.P.qlp_precision;
    rc = get_window(a.B, &a.win);
    if (rc) return rc;
    rc = get_window(a.tail_bs, &a.win_tail);
    if (rc) return rc;
    a.slots = reinterpret_cast<uint8_t*>(ws + pl.off_slots);
    a.slot_stride = pl.slot_stride;
    a.frame_bytes = reinterpret_cast<uint32_t*>(ws + pl.off_fbytes);
    a.info = reinterpret_cast<FrameInfo*>(d_info);
    a.pmax_full = max_porder_for(a.B, a.max_porder, 0);
    a.pmax_tail = max_porder_for(a.tail_bs, a.max_porder, 0);
    a.escale_full = 0.5 / (double)a.B;
    a.escale_tail = 0.5 / (double)a.tail_bs;
    {
        // frame header fields by frame number (as in encode_device_begin: tabulated on the host, cached on the device)
        void* dp = nullptr;
        const size_t ntab = (size_t)pl.nf * (nch == 2 ? 2 : 1);
        rc = get_scratch(9, ntab * sizeof(uint4) + 256, &dp);
        if (rc) return rc;
        if (ds_->c_nf != pl.nf || ds_->c_B != a.B || ds_->c_tail != a.tail_bs || ds_->c_nch != nch || ds_->c_dp != dp ||
            ds_->c_epoch != ds_->scratch_epoch) {
            ds_->h_hdr.resize(ntab);
            for (int64_t f = 0; f < pl.nf; ++f) {
                ds_->h_hdr[(size_t)f] = frame_header_entry((uint64_t)f, (f == pl.nf - 1) ? a.tail_bs : a.B, nch);
                if (nch == 2) ds_->h_hdr[(size_t)(pl.nf + f)] = frame_header_entry((uint64_t)f, (f == pl.nf - 1) ? a.tail_bs : a.B, nch, true);
            }
            FA_HIP_TRY(hipMemcpyAsync(dp, ds_->h_hdr.data(), ntab * sizeof(uint4), hipMemcpyHostToDevice, st));
            FA_HIP_TRY(hipStreamSynchronize(st));  // h_hdr is reused by the next call
            ds_->c_nf = pl.nf; ds_->c_B = a.B; ds_->c_tail = a.tail_bs; ds_->c_nch = nch; ds_->c_dp = dp; ds_->c_epoch = ds_->scratch_epoch;
        }
        a.hdr = reinterpret_cast<const uint4*>(dp);
    }
    FusedArgs p;
    std::memset(&p, 0, sizeof p);
    p.n_stream = n_stream; p.stream_size = stream_size; p.nframes = pl.nf; p.total_frames = pl.F;
    p.blob = d_bytes; p.capacity = capacity_bytes; p.hb = pl.hb;
    p.frame_bytes = a.frame_bytes;
    p.frame_abs = reinterpret_cast<int64_t*>(ws + pl.off_fabs);
    p.info = a.info;
    p.size_pub = reinterpret_cast<uint32_t*>(ws + pl.off_size);
    p.off_pub = reinterpret_cast<unsigned long long*>(ws + pl.off_off);
    // (the scanner's total lands next to the error word, inside the 256 zeroed bytes that hold the ticket: one copy brings both back)
    p.total = reinterpret_cast<int64_t*>(ws + pl.off_ticket + 16);
    p.ticket = reinterpret_cast<uint32_t*>(ws + pl.off_ticket);
    p.err = reinterpret_cast<int*>(ws + pl.off_ticket + 8);
    rc = get_crc_tab(&p.crc_tab);  // (K5's tables: the placement step is K5's per-frame copy)
    if (rc) return rc;
#ifdef FA_STAMPS
    {
        void* sp = nullptr;
        if (get_scratch(6, 512, &sp) == 0) {
            if (!ds_->stamps_zeroed) { (void)hipMemset(sp, 0, 512); ds_->stamps_zeroed = true; }
            p.stamps = reinterpret_cast<unsigned long long*>(sp);
            a.stamps = p.stamps;  // (the frame body's own phases: stamps[0..16], as in the slot kernel)
            (void)hipMemsetAsync(p.stamps + 28, 0, 8, st);  // (start time of the call's first workgroup)
        }
    }
#endif
    p.starts = d_starts;
    p.nbytes = d_nbytes;
    prof_begin(0, st);
    launch_encode_placed(st, a, p, nch, placed_grid(pl.F));
    prof_end(0, st);
    prof_end(3, st);
    // one wait: the error word and the total (the kernel has written the index and the stream headers itself)
    struct { int32_t err, pad; int64_t total; } back = {0, 0, 0};
    static_assert(sizeof back == 16, "error word at +8, total at +16 of the ticket block");
    FA_HIP_TRY(hipMemcpyAsync(&back, p.err, sizeof back, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    FA_HIP_TRY(hipGetLastError());
    const int h_err = back.err;
    const int64_t h_tot = back.total;
    if (h_err == 1 || (h_err == 0 && h_tot > capacity_bytes)) return FA_ERROR_ALLOC;  // the blob does not fit the caller's buffer
    if (h_err) {
        std::fprintf(stderr, "flacarray_hip: single-pass encode failed (flags %d: 1 = offset outside the buffer, 2 = a frame timed out waiting for its offset, 4 = the scanner timed out)\n", h_err);
        return FA_ERROR_ENCODE_PROCESS;
    }
    *h_total_bytes = h_tot;
    return FA_ERROR_NONE;
}

int fa_encode_i32_device(const int32_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                         int64_t workspace_bytes, unsigned char* d_bytes, int64_t capacity_bytes, int64_t* d_starts,
                         int64_t* d_nbytes, int64_t* h_total_bytes, int32_t* d_info, void* stream) {
    if (level > 8) return FA_ERROR_INVALID_LEVEL;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_ZERO_STREAMSIZE;
    if (slots_forced()) {  // diagnostic: the slot sequence into the same caller-provided buffer
        int rc = encode_device_begin(d_data, 1, n_stream, stream_size, level, d_workspace, workspace_bytes, d_starts, d_nbytes,
                                     h_total_bytes, d_info, stream);
        if (rc) return rc;
        if (*h_total_bytes > capacity_bytes) return FA_ERROR_ALLOC;
        return encode_device_finish(1, n_stream, stream_size, level, d_workspace, d_starts, d_bytes, stream);
    }
    // frames K3F does not cover (short blocks of levels 0-2, streams shorter than two frames, lengths that are not a multiple
    // of 4, unaligned rows): K3G
    if (!fused_geometry(n_stream, stream_size, level) || (reinterpret_cast<uintptr_t>(d_data) & 15) || placed_preferred(n_stream, stream_size))
        return placed_encode_run(d_data, 1, n_stream, stream_size, level, d_workspace, workspace_bytes, d_bytes, capacity_bytes, d_starts,
                                 d_nbytes, h_total_bytes, d_info, stream);
    return fused_encode_run(d_data, false, nullptr, nullptr, nullptr, n_stream, stream_size, level, d_workspace, workspace_bytes, d_bytes,
                            capacity_bytes, d_starts, d_nbytes, h_total_bytes, d_info, stream);
}

int fa_encode_f32_device(const float* d_data, int64_t n_stream, int64_t stream_size, uint32_t level, const float* d_quanta,
                         void* d_workspace, int64_t workspace_bytes, unsigned char* d_bytes, int64_t capacity_bytes,
                         int64_t* d_starts, int64_t* d_nbytes, float* d_offsets, float* d_gains, int64_t* h_total_bytes,
                         int32_t* d_info, void* stream) {
    if (level > 8) return FA_ERROR_INVALID_LEVEL;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_ZERO_STREAMSIZE;
    if (!d_offsets || !d_gains) return FA_ERROR_CONVERT_TYPE;
    // (other geometries: quantise with fa_float32_to_int32_device, then encode the integers)
    if (!fused_geometry(n_stream, stream_size, level, true) || (reinterpret_cast<uintptr_t>(d_data) & 15)) return FA_ERROR_ENCODE_INIT;
    return fused_encode_run(d_data, true, d_quanta, d_offsets, d_gains, n_stream, stream_size, level, d_workspace, workspace_bytes, d_bytes,
                            capacity_bytes, d_starts, d_nbytes, h_total_bytes, d_info, stream);
}

int fa_encode_i64_device(const int64_t* d_data, int64_t n_stream, int64_t stream_size, uint32_t level, void* d_workspace,
                         int64_t workspace_bytes, unsigned char* d_bytes, int64_t capacity_bytes, int64_t* d_starts,
                         int64_t* d_nbytes, int64_t* h_total_bytes, int32_t* d_info, void* stream) {
    if (level > 8) return FA_ERROR_INVALID_LEVEL;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_ZERO_STREAMSIZE;
    const int32_t* d32 = reinterpret_cast<const int32_t*>(d_data);
    if (slots_forced()) {
        int rc = encode_device_begin(d32, 2, n_stream, stream_size, level, d_workspace, workspace_bytes, d_starts, d_nbytes, h_total_bytes,
                                     d_info, stream);
        if (rc) return rc;
        if (*h_total_bytes > capacity_bytes) return FA_ERROR_ALLOC;
        return encode_device_finish(2, n_stream, stream_size, level, d_workspace, d_starts, d_bytes, stream);
    }
    return placed_encode_run(d32, 2, n_stream, stream_size, level, d_workspace, workspace_bytes, d_bytes, capacity_bytes, d_starts, d_nbytes,
                             h_total_bytes, d_info, stream);
}

int fa_decode_i32_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                         const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t first_sample,
                         int64_t last_sample, int32_t* d_out_i32, float* d_out_f32, const float* d_offsets,
                         const float* d_gains, void* stream, int verify) {
    FA_API_LOCK;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_DECODE_STREAMSIZE;
    if ((d_out_i32 == nullptr) == (d_out_f32 == nullptr)) return FA_ERROR_CONVERT_TYPE;
    if (d_out_f32 && (!d_offsets || !d_gains)) return FA_ERROR_CONVERT_TYPE;
    int64_t first_decode, n_decode;
    int rc = validate_range(stream_size, first_sample, last_sample, &first_decode, &n_decode);
    if (rc) return rc;
    return decode_device_impl(d_bytes, n_bytes, d_starts, d_nbytes, n_stream, stream_size, first_decode, n_decode, -1, nullptr,
                              nullptr, nullptr, nullptr, d_out_i32, d_out_f32, d_offsets, d_gains,
                              reinterpret_cast<hipStream_t>(stream), 1, nullptr, nullptr, nullptr, nullptr, nullptr, false, verify);
}

int fa_decode_slices_i32_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                                const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t n_slices,
                                const int64_t* slice_stream, const int64_t* slice_first, const int64_t* slice_count,
                                const int64_t* out_offset, int32_t* d_out_i32, float* d_out_f32,
                                const float* d_offsets, const float* d_gains, void* stream, int verify) {
    FA_API_LOCK;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_DECODE_STREAMSIZE;
    if (n_slices <= 0) return FA_ERROR_NONE;
    if ((d_out_i32 == nullptr) == (d_out_f32 == nullptr)) return FA_ERROR_CONVERT_TYPE;
    if (d_out_f32 && (!d_offsets || !d_gains)) return FA_ERROR_CONVERT_TYPE;
    return decode_device_impl(d_bytes, n_bytes, d_starts, d_nbytes, n_stream, stream_size, 0, 0, n_slices, slice_stream,
                              slice_first, slice_count, out_offset, d_out_i32, d_out_f32, d_offsets, d_gains,
                              reinterpret_cast<hipStream_t>(stream), 1, nullptr, nullptr, nullptr, nullptr, nullptr, false, verify);
}

int fa_decode_index_create(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts, const int64_t* d_nbytes,
                           int64_t n_stream, int64_t stream_size, int channels, void** index, void* stream) {
    FA_API_LOCK;
    if (!index) return FA_ERROR_ALLOC;
    *index = nullptr;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_DECODE_STREAMSIZE;
    if (channels != 1 && channels != 2) return FA_ERROR_CONVERT_TYPE;
    DecodeIndex* ix = new (std::nothrow) DecodeIndex();
    if (!ix) return FA_ERROR_ALLOC;
    (void)hipGetDevice(&ix->device);
    const int rc = decode_device_impl(d_bytes, n_bytes, d_starts, d_nbytes, n_stream, stream_size, 0, 0, -1, nullptr, nullptr, nullptr, nullptr,
                                      nullptr, nullptr, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream), channels, nullptr, nullptr,
                                      nullptr, nullptr, ix, true);
    if (rc) {
        if (ix->meta) (void)hipFree(ix->meta);
        if (ix->ftab) (void)hipFree(ix->ftab);
        delete ix;
        return rc;
    }
    *index = ix;
    return FA_ERROR_NONE;
}

void fa_decode_index_destroy(void* index) {
    DecodeIndex* ix = reinterpret_cast<DecodeIndex*>(index);
    if (!ix) return;
    // the index belongs to the device it was created on: free it there, whatever the caller's current device is
    int cur = -1;
    (void)hipGetDevice(&cur);
    const bool hop = (ix->device >= 0 && cur != ix->device);
    if (hop) (void)hipSetDevice(ix->device);
    struct Back { bool on; int dev; ~Back() { if (on) (void)hipSetDevice(dev); } } back{hop, cur};
    FA_API_LOCK_OR(return);
    if (ix->meta) (void)hipFree(ix->meta);
    if (ix->ftab) (void)hipFree(ix->ftab);
    if (ix->tasks) (void)hipFree(ix->tasks);
    delete ix;
}

int fa_decode_indexed(void* index, int64_t first_sample, int64_t last_sample, int64_t n_slices, const int64_t* slice_stream,
                      const int64_t* slice_first, const int64_t* slice_count, const int64_t* out_offset, void* d_out_int,
                      void* d_out_float, const void* d_offsets, const void* d_gains, void* stream, int verify) {
    FA_API_LOCK;
    DecodeIndex* ix = reinterpret_cast<DecodeIndex*>(index);
    if (!ix) return FA_ERROR_DECODE_INIT;
    {   // an index is used on the device that holds it (its tables, the store and this call's lock are that device's)
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != ix->device) return FA_ERROR_DEVICE;
    }
    if ((d_out_int == nullptr) == (d_out_float == nullptr)) return FA_ERROR_CONVERT_TYPE;
    if (d_out_float && (!d_offsets || !d_gains)) return FA_ERROR_CONVERT_TYPE;
    int64_t first_decode = 0, n_decode = 0;
    if (n_slices < 0) {
        const int rc = validate_range(ix->stream_size, first_sample, last_sample, &first_decode, &n_decode);
        if (rc) return rc;
    } else if (n_slices == 0) {
        return FA_ERROR_NONE;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (ix->nch == 1)
        return decode_device_impl(nullptr, 0, nullptr, nullptr, 0, 0, first_decode, n_decode, n_slices, slice_stream, slice_first, slice_count,
                                  out_offset, reinterpret_cast<int32_t*>(d_out_int), reinterpret_cast<float*>(d_out_float),
                                  reinterpret_cast<const float*>(d_offsets), reinterpret_cast<const float*>(d_gains), st, 1, nullptr, nullptr,
                                  nullptr, nullptr, ix, false, verify);
    return decode_device_impl(nullptr, 0, nullptr, nullptr, 0, 0, first_decode, n_decode, n_slices, slice_stream, slice_first, slice_count,
                              out_offset, nullptr, nullptr, nullptr, nullptr, st, 2, reinterpret_cast<int64_t*>(d_out_int),
                              reinterpret_cast<double*>(d_out_float), reinterpret_cast<const double*>(d_offsets),
                              reinterpret_cast<const double*>(d_gains), ix, false, verify);
}

void* fa_pinned_alloc(int64_t bytes) {
    if (bytes <= 0 || fa_device_count() <= 0) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void fa_pinned_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int fa_decode_indexed_host(void* index, int64_t n_slices, const int64_t* slice_stream, const int64_t* slice_first,
                           const int64_t* slice_count, const int64_t* out_offset, void* d_out_int, void* d_out_float,
                           const void* d_offsets, const void* d_gains, void* h_out, int64_t out_bytes, void* stream, int verify) {
    FA_API_LOCK;
    DecodeIndex* ix = reinterpret_cast<DecodeIndex*>(index);
    if (!ix || !h_out || out_bytes < 0) return FA_ERROR_DECODE_INIT;
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != ix->device) return FA_ERROR_DEVICE;
    }
    if ((d_out_int == nullptr) == (d_out_float == nullptr)) return FA_ERROR_CONVERT_TYPE;
    if (d_out_float && (!d_offsets || !d_gains)) return FA_ERROR_CONVERT_TYPE;
    if (n_slices <= 0) return FA_ERROR_NONE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool copied = false;
    int rc;
    if (ix->nch == 1)
        rc = decode_device_impl(nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, n_slices, slice_stream, slice_first, slice_count, out_offset,
                                reinterpret_cast<int32_t*>(d_out_int), reinterpret_cast<float*>(d_out_float),
                                reinterpret_cast<const float*>(d_offsets), reinterpret_cast<const float*>(d_gains), st, 1, nullptr, nullptr,
                                nullptr, nullptr, ix, false, verify, h_out, (size_t)out_bytes, &copied);
    else
        rc = decode_device_impl(nullptr, 0, nullptr, nullptr, 0, 0, 0, 0, n_slices, slice_stream, slice_first, slice_count, out_offset,
                                nullptr, nullptr, nullptr, nullptr, st, 2, reinterpret_cast<int64_t*>(d_out_int),
                                reinterpret_cast<double*>(d_out_float), reinterpret_cast<const double*>(d_offsets),
                                reinterpret_cast<const double*>(d_gains), ix, false, verify);
    if (rc) return rc;
    if (!copied && out_bytes > 0) {
        FA_HIP_TRY(hipMemcpyAsync(h_out, d_out_int ? d_out_int : d_out_float, (size_t)out_bytes, hipMemcpyDeviceToHost, st));
        FA_HIP_TRY(hipStreamSynchronize(st));
    }
    return FA_ERROR_NONE;
}

int fa_decode_i64_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                         const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t first_sample,
                         int64_t last_sample, int64_t* d_out_i64, double* d_out_f64, const double* d_offsets,
                         const double* d_gains, void* stream, int verify) {
    FA_API_LOCK;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_DECODE_STREAMSIZE;
    if ((d_out_i64 == nullptr) == (d_out_f64 == nullptr)) return FA_ERROR_CONVERT_TYPE;
    if (d_out_f64 && (!d_offsets || !d_gains)) return FA_ERROR_CONVERT_TYPE;
    int64_t first_decode, n_decode;
    int rc = validate_range(stream_size, first_sample, last_sample, &first_decode, &n_decode);
    if (rc) return rc;
    return decode_device_impl(d_bytes, n_bytes, d_starts, d_nbytes, n_stream, stream_size, first_decode, n_decode, -1, nullptr,
                              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream), 2,
                              d_out_i64, d_out_f64, d_offsets, d_gains, nullptr, false, verify);
}

int fa_decode_slices_i64_device(const unsigned char* d_bytes, int64_t n_bytes, const int64_t* d_starts,
                                const int64_t* d_nbytes, int64_t n_stream, int64_t stream_size, int64_t n_slices,
                                const int64_t* slice_stream, const int64_t* slice_first, const int64_t* slice_count,
                                const int64_t* out_offset, int64_t* d_out_i64, double* d_out_f64,
                                const double* d_offsets, const double* d_gains, void* stream, int verify) {
    FA_API_LOCK;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_DECODE_STREAMSIZE;
    if (n_slices <= 0) return FA_ERROR_NONE;
    if ((d_out_i64 == nullptr) == (d_out_f64 == nullptr)) return FA_ERROR_CONVERT_TYPE;
    if (d_out_f64 && (!d_offsets || !d_gains)) return FA_ERROR_CONVERT_TYPE;
    return decode_device_impl(d_bytes, n_bytes, d_starts, d_nbytes, n_stream, stream_size, 0, 0, n_slices, slice_stream,
                              slice_first, slice_count, out_offset, nullptr, nullptr, nullptr, nullptr,
                              reinterpret_cast<hipStream_t>(stream), 2, d_out_i64, d_out_f64, d_offsets, d_gains, nullptr, false, verify);
}

int fa_float32_to_int32_device(const float* d_input, int64_t n_stream, int64_t stream_size, const float* d_quanta,
                               int32_t* d_output, float* d_offsets, float* d_gains, void* stream) {
    FA_API_LOCK;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_ZERO_STREAMSIZE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    void* p = nullptr;
    int rc = get_scratch(4, 256, &p);
    if (rc) return rc;
    int* d_flags = reinterpret_cast<int*>(p);
    FA_HIP_TRY(hipMemsetAsync(d_flags, 0, 4, st));
    prof_begin(5, st);
    hipLaunchKernelGGL(float32_to_int32_kernel, dim3((unsigned)n_stream), dim3(1024), 0, st, d_input, stream_size, d_quanta,
                       d_output, d_offsets, d_gains, d_flags);
    prof_end(5, st);
    int h = 0;
    FA_HIP_TRY(hipMemcpyAsync(&h, d_flags, 4, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    FA_HIP_TRY(hipGetLastError());
    return (h & 1) ? FA_ERROR_NAN_INPUT : FA_ERROR_NONE;
}

int fa_int32_to_float32_device(const int32_t* d_input, int64_t n_stream, int64_t stream_size, const float* d_offsets,
                               const float* d_gains, float* d_output, void* stream) {
    FA_API_LOCK;
    if (n_stream <= 0 || stream_size <= 0) return FA_ERROR_NONE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t cps = (stream_size + kDequantChunk - 1) / kDequantChunk;
    hipLaunchKernelGGL(int32_to_float32_kernel, dim3((unsigned)(n_stream * cps)), dim3(256), 0, st, d_input, stream_size, cps,
                       d_offsets, d_gains, d_output);
    FA_HIP_TRY(hipGetLastError());
    return FA_ERROR_NONE;
}

int fa_float64_to_int64_device(const double* d_input, int64_t n_stream, int64_t stream_size, const double* d_quanta,
                               int64_t* d_output, double* d_offsets, double* d_gains, void* stream) {
    FA_API_LOCK;
    if (n_stream <= 0) return FA_ERROR_ZERO_NSTREAM;
    if (stream_size <= 0) return FA_ERROR_ZERO_STREAMSIZE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    void* p = nullptr;
    int rc = get_scratch(4, 256, &p);
    if (rc) return rc;
    int* d_flags = reinterpret_cast<int*>(p);
    FA_HIP_TRY(hipMemsetAsync(d_flags, 0, 4, st));
    hipLaunchKernelGGL(float64_to_int64_kernel, dim3((unsigned)n_stream), dim3(1024), 0, st, d_input, stream_size, d_quanta,
                       d_output, d_offsets, d_gains, d_flags);
    int h = 0;
    FA_HIP_TRY(hipMemcpyAsync(&h, d_flags, 4, hipMemcpyDeviceToHost, st));
    FA_HIP_TRY(hipStreamSynchronize(st));
    FA_HIP_TRY(hipGetLastError());
    return (h & 1) ? FA_ERROR_NAN_INPUT : FA_ERROR_NONE;
}

int fa_int64_to_float64_device(const int64_t* d_input, int64_t n_stream, int64_t stream_size, const double* d_offsets,
                               const double* d_gains, double* d_output, void* stream) {
    FA_API_LOCK;
    if (n_stream <= 0 || stream_size <= 0) return FA_ERROR_NONE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t cps = (stream_size + kDequantChunk - 1) / kDequantChunk;
    hipLaunchKernelGGL(int64_to_float64_kernel, dim3((unsigned)(n_stream * cps)), dim3(256), 0, st, d_input, stream_size, cps,
                       d_offsets, d_gains, d_output);
    FA_HIP_TRY(hipGetLastError());
    return FA_ERROR_NONE;
}

// ---------------------------------------------------------------------------------------------
// Host-pointer drop-ins (reference C ABI).  Data makes a PCIe round trip: the array is cut into chunks of
// ~256 MiB, a feeder thread uploads chunk c+1 on its own stream while this thread runs the kernels of chunk c and
// copies its result back (PCIe is full duplex), and the pages of freshly allocated host memory -- the malloc()'d blob
// of the encoder, the caller's output array of the decoder -- are populated by helper threads ahead of the copies
// (first-touch faults alone ran at 16 GB/s against the link's 56, tools/pcie_probe.py).  Device memory in use: two
// input chunks + one output chunk, whatever the array's size.
// ---------------------------------------------------------------------------------------------
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23  // Linux >= 5.14: fault pages in, writable, without touching their contents
#endif

// Populate [p, p + n) (the page-aligned inside of it) on a helper thread; join() before the memory is freed.  Contents
// are never written, so the helper may run beside DMA into the same range.  On kernels without MADV_POPULATE_WRITE the
// call fails and nothing happens (the copies then fault the pages in themselves, at 16 GB/s instead of the link's 56).
// Measured on the MI355X box (profiles/r03_host_abi.md): fresh anonymous memory populates at 19 GB/s (26 with
// transparent huge pages) whatever the number of threads -- that rate, not PCIe, bounds both host entry points -- and a
// populate call holds the address space's lock, which the runtime's pinning of the copy buffers and thread creation
// need too: one helper working in 128 KiB pieces lets the uploads through (4 helpers on 32 MiB pieces delayed the
// first upload by 40 ms).
static double host_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static bool host_trace() {  // FLACARRAY_HIP_HOST_TRACE=1: per-chunk timeline of the host entry points on stderr (tools/host_trace.py)
    static const bool on = std::getenv("FLACARRAY_HIP_HOST_TRACE") != nullptr;
    return on;
}
#define FA_HTRACE(...) do { if (host_trace()) std::fprintf(stderr, __VA_ARGS__); } while (0)

struct PagePopulator {
    std::vector<std::thread> th;
    void start(void* p, size_t n, int nthreads, bool huge) {
        const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
        uintptr_t a = ((uintptr_t)p + pg - 1) & ~(pg - 1), b = ((uintptr_t)p + n) & ~(pg - 1);
        if (b <= a || n < (64u << 20)) return;
        if (huge) {
            const uintptr_t hp = 2u << 20;
            const uintptr_t ha = (a + hp - 1) & ~(hp - 1), hb = b & ~(hp - 1);
            if (hb > ha) (void)madvise((void*)ha, hb - ha, MADV_HUGEPAGE);
        }
        const uintptr_t piece = 128u << 10;
        auto next = std::make_shared<std::atomic<uintptr_t>>(a);
        const double t0 = host_now();
        for (int t = 0; t < nthreads; ++t) try {
            th.emplace_back([next, b, piece, t0, t] {
                for (;;) {
                    const uintptr_t lo = next->fetch_add(piece);
                    if (lo >= b) break;
                    const uintptr_t hi = lo + piece < b ? lo + piece : b;
                    if (madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE) != 0) break;
                }
                FA_HTRACE("  populate thread %d done after %.2f ms\n", t, (host_now() - t0) * 1e3);
            });
        } catch (...) {  // no thread to be had: the copies fault the pages in themselves
            break;
        }
    }
    void join() {
        for (auto& t : th) t.join();
        th.clear();
    }
    ~PagePopulator() { join(); }
};

// Uploads: chunk c goes to slot c & 1 once chunk c - 2 has been consumed.  `upload(c, slot, stream)` issues the
// copies of one chunk; the thread waits for them and publishes the chunk.
struct Feeder {
    std::mutex mu;
    std::condition_variable cv;
    int64_t uploaded = 0, consumed = 0;  // chunks
    int err = FA_ERROR_NONE;
    bool stop = false;
    std::thread th;
    hipStream_t st = nullptr;
    int start(int dev, int64_t n_chunks, std::function<int(int64_t, int, hipStream_t)> upload, hipStream_t* cached) {
        if (!*cached && hipStreamCreateWithFlags(cached, hipStreamNonBlocking) != hipSuccess) return FA_ERROR_DEVICE;
        st = *cached;
        try {
        th = std::thread([this, dev, n_chunks, upload] {
            if (hipSetDevice(dev) != hipSuccess) { fail(FA_ERROR_DEVICE); return; }
            for (int64_t c = 0; c < n_chunks; ++c) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || consumed + 2 > c; });
                    if (stop) return;
                }
                const double t0 = host_now();
                int rc = upload(c, (int)(c & 1), st);
                if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = FA_ERROR_DEVICE;
                FA_HTRACE("  feeder: chunk %lld up in %.2f ms (started %.2f)\n", (long long)c, (host_now() - t0) * 1e3, t0 * 1e3);
                if (rc) { fail(rc); return; }
                { std::lock_guard<std::mutex> lk(mu); uploaded = c + 1; }
                cv.notify_all();
            }
        });
        } catch (...) {
            return FA_ERROR_ALLOC;  // (std::system_error: no thread)
        }
        return FA_ERROR_NONE;
    }
    void fail(int rc) {
        { std::lock_guard<std::mutex> lk(mu); err = rc; }
        cv.notify_all();
    }
    int wait_for(int64_t c) {  // chunk c is on the device (or the feeder failed)
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return err != FA_ERROR_NONE || uploaded > c; });
        return err;
    }
    void done_with(int64_t c) {
        { std::lock_guard<std::mutex> lk(mu); consumed = c + 1; }
        cv.notify_all();
    }
    void finish() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
        st = nullptr;  // (the stream stays with the device state)
    }
    ~Feeder() { finish(); }
};

static int64_t host_chunk_streams(int64_t n_stream, size_t bytes_per_stream, int64_t max_by_grid) {
    // ~256 MiB per chunk, at least 8 chunks for arrays that can afford them (the first upload and the last download are
    // not overlapped with anything), never more than the grid allows
    const size_t total = (size_t)n_stream * bytes_per_stream;
    size_t target = 256u << 20;
    if (total / 8 < target) target = total / 8 > (32u << 20) ? total / 8 : (32u << 20);
    if (const char* e = std::getenv("FLACARRAY_HIP_HOST_CHUNK_BYTES")) {  // tests: many small chunks through the pipeline
        const long long v = std::atoll(e);
        if (v > 0) target = (size_t)v;
    }
    int64_t chunk = (int64_t)(target / (bytes_per_stream ? bytes_per_stream : 1));
    if (chunk < 1) chunk = 1;
    if (chunk > n_stream) chunk = n_stream;
    if (chunk > max_by_grid) chunk = max_by_grid;
    if (chunk > (1 << 20)) chunk = 1 << 20;
    return chunk;
}

// data: int32 (nch 1), int64 (nch 2) or float32 (f32: quantised on the device -- fused into the encoder where the
// geometry allows, utils.c:160-243; quanta may be null; offsets / gains [n_stream] are outputs)
// f64 (nch == 2): the input is float64 and is quantised to int64 on the device (float64_to_int64, utils.c:245-327) in
// front of the two-channel encoder; quanta64 may be null; offsets64 / gains64 [n_stream] are outputs.
static int encode_host(const void* data_v, int nch, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
                       int64_t* starts, unsigned char** bytes, bool f32 = false, const float* quanta = nullptr,
                       float* offsets = nullptr, float* gains = nullptr, bool f64 = false, const double* quanta64 = nullptr,
                       double* offsets64 = nullptr, double* gains64 = nullptr) {
    FA_API_LOCK;
    const double t_enter = host_now();
    if (level > 8) return FA_ERROR_INVALID_LEVEL;        // compress.c:144-146
    if (n_stream == 0) return FA_ERROR_ZERO_NSTREAM;     // compress.c:147-149
    if (stream_size == 0) return FA_ERROR_ZERO_STREAMSIZE;  // compress.c:150-152
    *n_bytes = 0;
    *bytes = nullptr;
    for (int64_t i = 0; i < n_stream; ++i) starts[i] = 0;
    if (n_stream < 0 || stream_size < 0) return FA_ERROR_ZERO_NSTREAM;
    if (fa_device_count() <= 0) return FA_ERROR_DEVICE;
    EncodePlan one;
    int rc = make_plan(1, stream_size, level, &one, nch);
    if (rc) return rc;
    int dev = 0;
    FA_HIP_TRY(hipGetDevice(&dev));
    const unsigned char* data = reinterpret_cast<const unsigned char*>(data_v);
    const size_t stream_bytes = (size_t)stream_size * 4 * (size_t)nch;
    const int64_t chunk = host_chunk_streams(n_stream, stream_bytes, 0x7fffffffLL / one.nf);
    const int64_t n_chunks = (n_stream + chunk - 1) / chunk;
    const bool fused_f32 = f32 && fused_geometry(chunk, stream_size, level, true) &&
                           (n_stream % chunk == 0 || fused_geometry(n_stream % chunk, stream_size, level, true));

    // device buffers: two input slots, output, workspace, per-stream tables
    void *d_in2 = nullptr, *d_ws = nullptr, *d_aux = nullptr, *d_out = nullptr, *d_int = nullptr;
    const size_t in_b = (size_t)chunk * stream_bytes;
    const size_t in_slot = align_up(in_b, 256);
    if ((rc = get_scratch(0, 2 * in_slot + 256, &d_in2))) return rc;
    const int64_t wsb = single_pass_workspace_for(chunk, stream_size, level, nch);  // (the slot sequence's size when that is forced)
    if (wsb < 0) return FA_ERROR_ENCODE_PROCESS;
    if ((rc = get_scratch(5, (size_t)wsb, &d_ws))) return rc;
    if ((rc = get_scratch(4, (size_t)chunk * 40 + 1024, &d_aux))) return rc;
    int64_t* d_starts = reinterpret_cast<int64_t*>(d_aux);
    int64_t* d_nb = d_starts + chunk;
    float* d_q = reinterpret_cast<float*>(d_nb + chunk);
    float* d_off = d_q + chunk;
    float* d_gain = d_off + chunk;
    double* d_q64 = reinterpret_cast<double*>(d_nb + chunk);  // (the float64 form uses the same region: three doubles per stream)
    double* d_off64 = d_q64 + chunk;
    double* d_gain64 = d_off64 + chunk;
    const int64_t cap_chunk = capacity_bytes_for(chunk, stream_size, level, nch);
    if ((rc = get_scratch(3, (size_t)cap_chunk + 256, &d_out))) return rc;
    if (((f32 && !fused_f32) || f64) && (rc = get_scratch(11, in_b + 256, &d_int))) return rc;

    // the blob: worst case reserved (address space only), populated ahead of the copies, trimmed at the end
    const int64_t cap_total = n_stream * one.nf * (int64_t)kSlotBytes * nch + n_stream * stream_header_bytes(one.nf) + 64;
    unsigned char* blob = reinterpret_cast<unsigned char*>(std::malloc((size_t)cap_total));
    bool reserved = (blob != nullptr);
    if (!reserved) {  // no overcommit: fall back to growing the blob chunk by chunk
        blob = reinterpret_cast<unsigned char*>(std::malloc(1));
        if (!blob) return FA_ERROR_ALLOC;
    }
    size_t blob_cap = reserved ? (size_t)cap_total : 1;
    FA_HTRACE("encode: buffers %.2f ms\n", (host_now() - t_enter) * 1e3);
    PagePopulator pop;
    // (what a typical array compresses to; the rest of the reservation stays untouched address space)
    if (reserved) pop.start(blob, std::min<size_t>((size_t)cap_total, (size_t)n_stream * stream_bytes / 8 * 5), 1, true);

    FA_HTRACE("encode: populate started %.2f ms\n", (host_now() - t_enter) * 1e3);
    Feeder feed;
    int err = FA_ERROR_NONE;
    if (n_chunks > 1) {
        err = feed.start(dev, n_chunks, [=](int64_t c, int slot, hipStream_t st) {
            const int64_t s0 = c * chunk, ns = std::min(chunk, n_stream - s0);
            if (hipMemcpyAsync(reinterpret_cast<char*>(d_in2) + (size_t)slot * in_slot, data + (size_t)s0 * stream_bytes, (size_t)ns * stream_bytes,
                               hipMemcpyHostToDevice, st) != hipSuccess) return (int)FA_ERROR_DEVICE;
            return (int)FA_ERROR_NONE;
        }, &ds_->feed_stream);
    }
    int64_t running = 0;
    FA_HTRACE("encode: set-up %.2f ms (entered at %.2f)\n", (host_now() - t_enter) * 1e3, t_enter * 1e3);
    for (int64_t c = 0; c < n_chunks && !err; ++c) {
        const int64_t s0 = c * chunk, ns = std::min(chunk, n_stream - s0);
        void* d_in = reinterpret_cast<char*>(d_in2) + (size_t)(c & 1) * in_slot;
        const double tw0 = host_now();
        if (n_chunks > 1) {
            if ((err = feed.wait_for(c))) break;
        } else if (hipMemcpy(d_in, data, (size_t)ns * stream_bytes, hipMemcpyHostToDevice) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        const double tw1 = host_now();
        int64_t total = 0;
        if (f32) {
            if (quanta && hipMemcpy(d_q, quanta + s0, (size_t)ns * 4, hipMemcpyHostToDevice) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
            const int32_t* ints = nullptr;
            if (fused_f32) {
                err = fa_encode_f32_device(reinterpret_cast<const float*>(d_in), ns, stream_size, level, quanta ? d_q : nullptr, d_ws, wsb,
                                           reinterpret_cast<unsigned char*>(d_out), cap_chunk, d_starts, d_nb, d_off, d_gain, &total, nullptr, nullptr);
            } else {
                err = fa_float32_to_int32_device(reinterpret_cast<const float*>(d_in), ns, stream_size, quanta ? d_q : nullptr,
                                                 reinterpret_cast<int32_t*>(d_int), d_off, d_gain, nullptr);
                ints = reinterpret_cast<const int32_t*>(d_int);
                if (!err) err = fa_encode_i32_device(ints, ns, stream_size, level, d_ws, wsb, reinterpret_cast<unsigned char*>(d_out), cap_chunk,
                                                     d_starts, d_nb, &total, nullptr, nullptr);
            }
            if (err) break;
            if (hipMemcpy(offsets + s0, d_off, (size_t)ns * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(gains + s0, d_gain, (size_t)ns * 4, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        } else if (nch == 1) {
            err = fa_encode_i32_device(reinterpret_cast<const int32_t*>(d_in), ns, stream_size, level, d_ws, wsb,
                                       reinterpret_cast<unsigned char*>(d_out), cap_chunk, d_starts, d_nb, &total, nullptr, nullptr);
            if (err) break;
        } else {
            const void* src64 = d_in;
            if (f64) {
                if (quanta64 && hipMemcpy(d_q64, quanta64 + s0, (size_t)ns * 8, hipMemcpyHostToDevice) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
                err = fa_float64_to_int64_device(reinterpret_cast<const double*>(d_in), ns, stream_size, quanta64 ? d_q64 : nullptr,
                                                 reinterpret_cast<int64_t*>(d_int), d_off64, d_gain64, nullptr);
                if (err) break;
                if (hipMemcpy(offsets64 + s0, d_off64, (size_t)ns * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(gains64 + s0, d_gain64, (size_t)ns * 8, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
                src64 = d_int;
            }
            err = fa_encode_i64_device(reinterpret_cast<const int64_t*>(src64), ns, stream_size, level, d_ws, wsb, reinterpret_cast<unsigned char*>(d_out),
                                       cap_chunk, d_starts, d_nb, &total, nullptr, nullptr);
            if (err) break;
        }
        if (n_chunks > 1) {
            // the kernels are done with the input slot (the encode calls end with a stream synchronisation)
            feed.done_with(c);
        }
        if ((size_t)(running + total) > blob_cap) {  // (only without the reservation)
            pop.join();
            unsigned char* nb2 = reinterpret_cast<unsigned char*>(std::realloc(blob, (size_t)(running + total)));
            if (!nb2) { err = FA_ERROR_ALLOC; break; }
            blob = nb2;
            blob_cap = (size_t)(running + total);
        }
        const double tw2 = host_now();
        if (hipMemcpy(blob + running, d_out, (size_t)total, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        if (hipMemcpy(starts + s0, d_starts, (size_t)ns * 8, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        FA_HTRACE("encode chunk %lld: wait %.2f ms, kernels %.2f ms, download %.2f ms (%.1f MB) at %.2f\n", (long long)c, (tw1 - tw0) * 1e3,
                  (tw2 - tw1) * 1e3, (host_now() - tw2) * 1e3, total / 1e6, tw0 * 1e3);
        for (int64_t i = 0; i < ns; ++i) starts[s0 + i] += running;
        running += total;
    }
    const double tf0 = host_now();
    feed.finish();
    pop.join();
    FA_HTRACE("encode: joins %.2f ms\n", (host_now() - tf0) * 1e3);
    if (err) {
        std::free(blob);
        for (int64_t i = 0; i < n_stream; ++i) starts[i] = 0;
        return err;
    }
    // trim the reservation to the bytes written (a shrinking realloc of an mmap'd block returns its tail to the system)
    const double tr0 = host_now();
    unsigned char* fit = reinterpret_cast<unsigned char*>(std::realloc(blob, running > 0 ? (size_t)running : 1));
    FA_HTRACE("encode: trim %.2f ms, whole call %.2f ms\n", (host_now() - tr0) * 1e3, (host_now() - t_enter) * 1e3);
    *bytes = fit ? fit : blob;
    *n_bytes = running;
    return FA_ERROR_NONE;
}

int fa_encode_f32_host(const float* data, int64_t n_stream, int64_t stream_size, uint32_t level, const float* quanta, int64_t* n_bytes,
                       int64_t* starts, unsigned char** bytes, float* offsets, float* gains) {
    if (!offsets || !gains) return FA_ERROR_CONVERT_TYPE;
    return encode_host(data, 1, n_stream, stream_size, level, n_bytes, starts, bytes, true, quanta, offsets, gains);
}

int fa_encode_f64_host(const double* data, int64_t n_stream, int64_t stream_size, uint32_t level, const double* quanta, int64_t* n_bytes,
                       int64_t* starts, unsigned char** bytes, double* offsets, double* gains) {
    if (!offsets || !gains) return FA_ERROR_CONVERT_TYPE;
    return encode_host(data, 2, n_stream, stream_size, level, n_bytes, starts, bytes, false, nullptr, nullptr, nullptr, true, quanta, offsets, gains);
}

int encode_i32(int32_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
               int64_t* starts, unsigned char** bytes) {
    return encode_host(data, 1, n_stream, stream_size, level, n_bytes, starts, bytes);
}

int encode_i32_threaded(int32_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
                        int64_t* starts, unsigned char** bytes) {
    return encode_host(data, 1, n_stream, stream_size, level, n_bytes, starts, bytes);
}

int encode_i64(int64_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
               int64_t* starts, unsigned char** bytes) {
    return encode_host(reinterpret_cast<const int32_t*>(data), 2, n_stream, stream_size, level, n_bytes, starts, bytes);
}

int encode_i64_threaded(int64_t* const data, int64_t n_stream, int64_t stream_size, uint32_t level, int64_t* n_bytes,
                        int64_t* starts, unsigned char** bytes) {
    return encode_host(reinterpret_cast<const int32_t*>(data), 2, n_stream, stream_size, level, n_bytes, starts, bytes);
}

// One chunk of streams as the decoder wants it: the byte ranges of its streams, sorted and merged where they touch (or
// nearly touch), packed into one device buffer.  Only those ranges are uploaded -- a keep mask that selects a few
// streams of a large store moves the bytes of those streams, not everything between the first and the last.
struct ChunkRanges {
    struct Piece { int64_t src, len, dst; };
    std::vector<Piece> pieces;
    std::vector<int64_t> rel;  // start of every stream of the chunk inside the packed buffer
    int64_t packed = 0;
    bool ok = true;
    void build(const int64_t* starts, const int64_t* nbytes, int64_t s0, int64_t ns) {
        pieces.clear();
        rel.assign((size_t)ns, 0);
        packed = 0;
        ok = true;
        std::vector<int64_t> order((size_t)ns);
        for (int64_t i = 0; i < ns; ++i) {
            order[(size_t)i] = i;
            // the C signature carries no blob length: negative or overflowing entries are all that can be refused here
            if (starts[s0 + i] < 0 || nbytes[s0 + i] < 0 || starts[s0 + i] > INT64_MAX - nbytes[s0 + i]) ok = false;
        }
        if (!ok) return;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return starts[s0 + a] < starts[s0 + b]; });
        int64_t lo = -1, hi = -1;
        auto flush = [&]() {
            if (lo < 0) return;
            pieces.push_back({lo, hi - lo, packed});
            packed += (hi - lo + 255) / 256 * 256;  // every piece starts 256-byte aligned
        };
        std::vector<int64_t> piece_of((size_t)ns);
        for (int64_t k = 0; k < ns; ++k) {
            const int64_t i = order[(size_t)k], a = starts[s0 + i], b = a + nbytes[s0 + i];
            if (lo >= 0 && a <= hi + 4096) {
                if (b > hi) hi = b;
            } else {
                flush();
                lo = a; hi = b;
            }
            piece_of[(size_t)i] = (int64_t)pieces.size();  // (index of the piece being built)
        }
        flush();
        if (pieces.size() > 2048) {  // too scattered for one copy per piece: everything between the first and the last byte
            const int64_t a = pieces.front().src, b = pieces.back().src + pieces.back().len;
            pieces.assign(1, {a, b - a, 0});
            packed = b - a;
            for (int64_t i = 0; i < ns; ++i) rel[(size_t)i] = starts[s0 + i] - a;
            return;
        }
        for (int64_t i = 0; i < ns; ++i) {
            const Piece& pc = pieces[(size_t)piece_of[(size_t)i]];
            rel[(size_t)i] = pc.dst + (starts[s0 + i] - pc.src);
        }
    }
};

// offsets / gains (both or neither; float for one channel, double for two): the int -> float restore (utils.c:329-368) is
// fused into the decoder's store and `data_v` receives float32 / float64 -- the integers never cross PCIe.
static int decode_host(const unsigned char* bytes, const int64_t* starts, const int64_t* nbytes, int64_t n_stream,
                       int64_t stream_size, int64_t first_sample, int64_t last_sample, void* data_v, int nch,
                       const void* offsets = nullptr, const void* gains = nullptr) {
    FA_API_LOCK;
    const double t_enter = host_now();
    const size_t esz = 4 * (size_t)nch;  // bytes per decoded sample
    unsigned char* data = reinterpret_cast<unsigned char*>(data_v);
    int64_t first_decode, n_decode;
    int rc = validate_range(stream_size, first_sample, last_sample, &first_decode, &n_decode);  // decompress.c:209-222
    if (rc) return rc;
    if (n_stream <= 0) return FA_ERROR_NONE;
    if (fa_device_count() <= 0) return FA_ERROR_DEVICE;
    int dev = 0;
    FA_HIP_TRY(hipGetDevice(&dev));
    // The reference's decoder always checks the frame CRC-16 (libFLAC reports a mismatch through the error callback,
    // decompress.c:104-121); on this entry point the check is on unless FLACARRAY_HIP_HOST_VERIFY=0 -- the bytes
    // crossed PCIe anyway, one more read of them in HBM is in the noise.
    const char* hv = std::getenv("FLACARRAY_HIP_HOST_VERIFY");
    const int host_verify = (hv && hv[0] == '0') ? 0 : 1;
    const size_t row_bytes = (size_t)n_decode * esz;
    // Chunks are bounded on BOTH sides: by decoded bytes (~256 MiB of output) and by the compressed bytes of their
    // streams (~256 MiB of input) -- a short sample range over a large store (arr[:, 0:100]) has tiny rows and would
    // otherwise put the whole store into one chunk.  Device memory in use: two input slots (one when there is a single
    // chunk) + one output chunk, whatever the store's size.
    const int64_t chunk = host_chunk_streams(n_stream, row_bytes, 0x7fffffffLL);  // (streams per chunk by output bytes)
    int64_t in_target = 256ll << 20;
    if (const char* e = std::getenv("FLACARRAY_HIP_HOST_CHUNK_BYTES")) {
        const long long v = std::atoll(e);
        if (v > 0) in_target = v;
    }
    std::vector<int64_t> cb;  // chunk c = streams [cb[c], cb[c + 1])
    cb.push_back(0);
    {
        int64_t in_sum = 0;
        for (int64_t s = 0; s < n_stream; ++s) {
            const int64_t nbs = nbytes[s] > 0 ? nbytes[s] : 0;
            if (s > cb.back() && (s - cb.back() >= chunk || in_sum + nbs > in_target)) { cb.push_back(s); in_sum = 0; }
            in_sum += nbs;
        }
        cb.push_back(n_stream);
    }
    const int64_t n_chunks = (int64_t)cb.size() - 1;

    // byte ranges of every chunk (host side, cheap), and the largest packed size: the upload slots are that large
    std::vector<ChunkRanges> cr((size_t)n_chunks);
    int64_t max_packed = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t s0 = cb[(size_t)c], ns = cb[(size_t)c + 1] - s0;
        cr[(size_t)c].build(starts, nbytes, s0, ns);
        if (!cr[(size_t)c].ok || cr[(size_t)c].packed <= 0) return FA_ERROR_DECODE_INIT;
        max_packed = std::max(max_packed, cr[(size_t)c].packed);
    }
    void *d_blob2 = nullptr, *d_aux2 = nullptr, *d_out = nullptr;
    const size_t blob_slot = align_up((size_t)max_packed + 256, 256);
    const size_t aux_slot = align_up((size_t)chunk * 32 + 512, 256);  // starts, nbytes, (offsets, gains: up to 8 B each)
    int err = FA_ERROR_NONE;
    if ((err = get_scratch(0, (n_chunks > 1 ? 2 : 1) * blob_slot, &d_blob2))) return err;
    if ((err = get_scratch(4, 2 * aux_slot, &d_aux2))) return err;
    if ((err = get_scratch(5, (size_t)chunk * row_bytes * (nch == 2 ? 1 : 1) + 256, &d_out))) return err;

    // the caller's output array is usually fresh memory (np.empty): populate its pages beside the first upload
    FA_HTRACE("decode: buffers %.2f ms\n", (host_now() - t_enter) * 1e3);
    PagePopulator pop;
    pop.start(data, (size_t)n_stream * row_bytes, 1, false);
    FA_HTRACE("decode: populate started %.2f ms\n", (host_now() - t_enter) * 1e3);

    auto upload = [&, bytes, nbytes](int64_t c, int slot, hipStream_t st) -> int {
        const int64_t s0 = cb[(size_t)c], ns = cb[(size_t)c + 1] - s0;
        const ChunkRanges& r = cr[(size_t)c];
        char* db = reinterpret_cast<char*>(d_blob2) + (size_t)slot * blob_slot;
        for (const auto& pc : r.pieces)
            if (hipMemcpyAsync(db + pc.dst, bytes + pc.src, (size_t)pc.len, hipMemcpyHostToDevice, st) != hipSuccess) return FA_ERROR_DEVICE;
        char* da = reinterpret_cast<char*>(d_aux2) + (size_t)slot * aux_slot;
        if (hipMemcpyAsync(da, r.rel.data(), (size_t)ns * 8, hipMemcpyHostToDevice, st) != hipSuccess) return FA_ERROR_DEVICE;
        if (hipMemcpyAsync(da + (size_t)chunk * 8, nbytes + s0, (size_t)ns * 8, hipMemcpyHostToDevice, st) != hipSuccess) return FA_ERROR_DEVICE;
        if (offsets) {
            const size_t fsz = (nch == 2) ? 8 : 4;
            if (hipMemcpyAsync(da + (size_t)chunk * 16, reinterpret_cast<const char*>(offsets) + (size_t)s0 * fsz, (size_t)ns * fsz, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(da + (size_t)chunk * 24, reinterpret_cast<const char*>(gains) + (size_t)s0 * fsz, (size_t)ns * fsz, hipMemcpyHostToDevice, st) != hipSuccess)
                return FA_ERROR_DEVICE;
        }
        return FA_ERROR_NONE;
    };
    Feeder feed;
    if (n_chunks > 1) err = feed.start(dev, n_chunks, upload, &ds_->feed_stream);
    FA_HTRACE("decode: set-up %.2f ms (entered at %.2f)\n", (host_now() - t_enter) * 1e3, t_enter * 1e3);
    for (int64_t c = 0; c < n_chunks && !err; ++c) {
        const int64_t s0 = cb[(size_t)c], ns = cb[(size_t)c + 1] - s0;
        const int slot = (int)(c & 1);
        if (n_chunks > 1) {
            if ((err = feed.wait_for(c))) break;
        } else {
            if ((err = upload(0, 0, nullptr))) break;
            if (hipStreamSynchronize(nullptr) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        }
        const unsigned char* db = reinterpret_cast<const unsigned char*>(d_blob2) + (size_t)slot * blob_slot;
        const int64_t* d_starts = reinterpret_cast<const int64_t*>(reinterpret_cast<char*>(d_aux2) + (size_t)slot * aux_slot);
        const int64_t* d_nb = d_starts + chunk;
        const int64_t packed = cr[(size_t)c].packed;
        const char* d_fo = reinterpret_cast<const char*>(d_starts) + (size_t)chunk * 16;  // offsets, gains of the chunk (if any)
        const char* d_fg = reinterpret_cast<const char*>(d_starts) + (size_t)chunk * 24;
        if (nch == 1)
            err = decode_device_impl(db, packed, d_starts, d_nb, ns, stream_size, first_decode, n_decode, -1, nullptr, nullptr, nullptr, nullptr,
                                     offsets ? nullptr : reinterpret_cast<int32_t*>(d_out), offsets ? reinterpret_cast<float*>(d_out) : nullptr,
                                     offsets ? reinterpret_cast<const float*>(d_fo) : nullptr, offsets ? reinterpret_cast<const float*>(d_fg) : nullptr,
                                     nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, false, host_verify);
        else
            err = decode_device_impl(db, packed, d_starts, d_nb, ns, stream_size, first_decode, n_decode, -1, nullptr, nullptr, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, 2, offsets ? nullptr : reinterpret_cast<int64_t*>(d_out),
                                     offsets ? reinterpret_cast<double*>(d_out) : nullptr, offsets ? reinterpret_cast<const double*>(d_fo) : nullptr,
                                     offsets ? reinterpret_cast<const double*>(d_fg) : nullptr, nullptr, false, host_verify);
        if (n_chunks > 1) feed.done_with(c);  // (the decode call ends with a stream synchronisation: the slot is free)
        if (err) break;
        const double td0 = host_now();
        if (hipMemcpy(data + (size_t)s0 * row_bytes, d_out, (size_t)ns * row_bytes, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        FA_HTRACE("decode chunk %lld: download %.2f ms at %.2f\n", (long long)c, (host_now() - td0) * 1e3, td0 * 1e3);
    }
    const double tj0 = host_now();
    feed.finish();
    pop.join();
    FA_HTRACE("decode: joins %.2f ms, whole call %.2f ms\n", (host_now() - tj0) * 1e3, (host_now() - t_enter) * 1e3);
    return err;
}

int decode_i32(unsigned char* const bytes, int64_t* const starts, int64_t* const nbytes, int64_t n_stream,
               int64_t stream_size, int64_t first_sample, int64_t last_sample, int32_t* data, bool use_threads) {
    (void)use_threads;
    return decode_host(bytes, starts, nbytes, n_stream, stream_size, first_sample, last_sample, data, 1);
}

int decode_i64(unsigned char* const bytes, int64_t* const starts, int64_t* const nbytes, int64_t n_stream,
               int64_t stream_size, int64_t first_sample, int64_t last_sample, int64_t* data, bool use_threads) {
    (void)use_threads;
    return decode_host(bytes, starts, nbytes, n_stream, stream_size, first_sample, last_sample, data, 2);
}

int fa_decode_f32_host(const unsigned char* bytes, const int64_t* starts, const int64_t* nbytes, int64_t n_stream, int64_t stream_size,
                       int64_t first_sample, int64_t last_sample, const float* offsets, const float* gains, float* data) {
    if (!offsets || !gains) return FA_ERROR_CONVERT_TYPE;
    return decode_host(bytes, starts, nbytes, n_stream, stream_size, first_sample, last_sample, data, 1, offsets, gains);
}

int fa_decode_f64_host(const unsigned char* bytes, const int64_t* starts, const int64_t* nbytes, int64_t n_stream, int64_t stream_size,
                       int64_t first_sample, int64_t last_sample, const double* offsets, const double* gains, double* data) {
    if (!offsets || !gains) return FA_ERROR_CONVERT_TYPE;
    return decode_host(bytes, starts, nbytes, n_stream, stream_size, first_sample, last_sample, data, 2, offsets, gains);
}

int float32_to_int32(float const* input, int64_t n_stream, int64_t stream_size, float const* quanta, int32_t* output,
                     float* offsets, float* gains) {
    FA_API_LOCK;
    if (n_stream <= 0 || stream_size <= 0) return FA_ERROR_NONE;
    if (fa_device_count() <= 0) return FA_ERROR_DEVICE;
    size_t free_b = 0, total_b = 0;
    FA_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int64_t chunk = (int64_t)((free_b / 10 * 8) / ((size_t)stream_size * 8 + 64));
    if (chunk < 1) chunk = 1;
    if (chunk > n_stream) chunk = n_stream;
    int err = FA_ERROR_NONE;
    for (int64_t s0 = 0; s0 < n_stream && !err; s0 += chunk) {
        const int64_t ns = (n_stream - s0 < chunk) ? (n_stream - s0) : chunk;
        const size_t nb = (size_t)ns * (size_t)stream_size * 4;
        void *d_in = nullptr, *d_out = nullptr, *d_aux = nullptr;
        if ((err = get_scratch(0, nb, &d_in))) break;
        if ((err = get_scratch(5, nb, &d_out))) break;
        if ((err = get_scratch(3, (size_t)ns * 12 + 768, &d_aux))) break;
        float* d_q = reinterpret_cast<float*>(d_aux);
        float* d_off = d_q + ns;
        float* d_gain = d_off + ns;
        if (hipMemcpy(d_in, input + s0 * stream_size, nb, hipMemcpyHostToDevice) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        if (quanta && hipMemcpy(d_q, quanta + s0, (size_t)ns * 4, hipMemcpyHostToDevice) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        err = fa_float32_to_int32_device(reinterpret_cast<const float*>(d_in), ns, stream_size, quanta ? d_q : nullptr,
                                         reinterpret_cast<int32_t*>(d_out), d_off, d_gain, nullptr);
        if (err) break;
        if (hipMemcpy(output + s0 * stream_size, d_out, nb, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        if (hipMemcpy(offsets + s0, d_off, (size_t)ns * 4, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        if (hipMemcpy(gains + s0, d_gain, (size_t)ns * 4, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
    }
    return err;
}

void int32_to_float32(int32_t const* input, int64_t n_stream, int64_t stream_size, float const* offsets,
                      float const* gains, float* output) {
    FA_API_LOCK_OR(fatal_device("int32_to_float32", "hipGetDevice"));
    if (n_stream <= 0 || stream_size <= 0) return;
    if (fa_device_count() <= 0) fatal_device("int32_to_float32", "hipGetDeviceCount (no HIP device)");
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) fatal_device("int32_to_float32", "hipMemGetInfo");
    int64_t chunk = (int64_t)((free_b / 10 * 8) / ((size_t)stream_size * 8 + 64));
    if (chunk < 1) chunk = 1;
    if (chunk > n_stream) chunk = n_stream;
    for (int64_t s0 = 0; s0 < n_stream; s0 += chunk) {
        const int64_t ns = (n_stream - s0 < chunk) ? (n_stream - s0) : chunk;
        const size_t nb = (size_t)ns * (size_t)stream_size * 4;
        void *d_in = nullptr, *d_out = nullptr, *d_aux = nullptr;
        const char* fn = "int32_to_float32";
        if (get_scratch(0, nb, &d_in) || get_scratch(5, nb, &d_out) || get_scratch(3, (size_t)ns * 8 + 512, &d_aux))
            fatal_device(fn, "hipMalloc of the staging buffers");
        float* d_off = reinterpret_cast<float*>(d_aux);
        float* d_gain = d_off + ns;
        if (hipMemcpy(d_in, input + s0 * stream_size, nb, hipMemcpyHostToDevice) != hipSuccess) fatal_device(fn, "hipMemcpy(input, H2D)");
        if (hipMemcpy(d_off, offsets + s0, (size_t)ns * 4, hipMemcpyHostToDevice) != hipSuccess) fatal_device(fn, "hipMemcpy(offsets, H2D)");
        if (hipMemcpy(d_gain, gains + s0, (size_t)ns * 4, hipMemcpyHostToDevice) != hipSuccess) fatal_device(fn, "hipMemcpy(gains, H2D)");
        if (fa_int32_to_float32_device(reinterpret_cast<const int32_t*>(d_in), ns, stream_size, d_off, d_gain,
                                       reinterpret_cast<float*>(d_out), nullptr) != FA_ERROR_NONE)
            fatal_device(fn, "int32_to_float32_kernel launch");
        if (hipMemcpy(output + s0 * stream_size, d_out, nb, hipMemcpyDeviceToHost) != hipSuccess) fatal_device(fn, "hipMemcpy(output, D2H)");
    }
}

int float64_to_int64(double const* input, int64_t n_stream, int64_t stream_size, double const* quanta, int64_t* output,
                     double* offsets, double* gains) {
    FA_API_LOCK;
    if (n_stream <= 0 || stream_size <= 0) return FA_ERROR_NONE;
    if (fa_device_count() <= 0) return FA_ERROR_DEVICE;
    size_t free_b = 0, total_b = 0;
    FA_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int64_t chunk = (int64_t)((free_b / 10 * 8) / ((size_t)stream_size * 16 + 64));
    if (chunk < 1) chunk = 1;
    if (chunk > n_stream) chunk = n_stream;
    int err = FA_ERROR_NONE;
    for (int64_t s0 = 0; s0 < n_stream && !err; s0 += chunk) {
        const int64_t ns = (n_stream - s0 < chunk) ? (n_stream - s0) : chunk;
        const size_t nb = (size_t)ns * (size_t)stream_size * 8;
        void *d_in = nullptr, *d_out = nullptr, *d_aux = nullptr;
        if ((err = get_scratch(0, nb, &d_in))) break;
        if ((err = get_scratch(5, nb, &d_out))) break;
        if ((err = get_scratch(3, (size_t)ns * 24 + 768, &d_aux))) break;
        double* d_q = reinterpret_cast<double*>(d_aux);
        double* d_off = d_q + ns;
        double* d_gain = d_off + ns;
        if (hipMemcpy(d_in, input + s0 * stream_size, nb, hipMemcpyHostToDevice) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        if (quanta && hipMemcpy(d_q, quanta + s0, (size_t)ns * 8, hipMemcpyHostToDevice) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        err = fa_float64_to_int64_device(reinterpret_cast<const double*>(d_in), ns, stream_size, quanta ? d_q : nullptr,
                                         reinterpret_cast<int64_t*>(d_out), d_off, d_gain, nullptr);
        if (err) break;
        if (hipMemcpy(output + s0 * stream_size, d_out, nb, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        if (hipMemcpy(offsets + s0, d_off, (size_t)ns * 8, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
        if (hipMemcpy(gains + s0, d_gain, (size_t)ns * 8, hipMemcpyDeviceToHost) != hipSuccess) { err = FA_ERROR_DEVICE; break; }
    }
    return err;
}

void int64_to_float64(int64_t const* input, int64_t n_stream, int64_t stream_size, double const* offsets,
                      double const* gains, double* output) {
    FA_API_LOCK_OR(fatal_device("int64_to_float64", "hipGetDevice"));
    if (n_stream <= 0 || stream_size <= 0) return;
    if (fa_device_count() <= 0) fatal_device("int64_to_float64", "hipGetDeviceCount (no HIP device)");
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) fatal_device("int64_to_float64", "hipMemGetInfo");
    int64_t chunk = (int64_t)((free_b / 10 * 8) / ((size_t)stream_size * 16 + 64));
    if (chunk < 1) chunk = 1;
    if (chunk > n_stream) chunk = n_stream;
    for (int64_t s0 = 0; s0 < n_stream; s0 += chunk) {
        const int64_t ns = (n_stream - s0 < chunk) ? (n_stream - s0) : chunk;
        const size_t nb = (size_t)ns * (size_t)stream_size * 8;
        void *d_in = nullptr, *d_out = nullptr, *d_aux = nullptr;
        const char* fn = "int64_to_float64";
        if (get_scratch(0, nb, &d_in) || get_scratch(5, nb, &d_out) || get_scratch(3, (size_t)ns * 16 + 512, &d_aux))
            fatal_device(fn, "hipMalloc of the staging buffers");
        double* d_off = reinterpret_cast<double*>(d_aux);
        double* d_gain = d_off + ns;
        if (hipMemcpy(d_in, input + s0 * stream_size, nb, hipMemcpyHostToDevice) != hipSuccess) fatal_device(fn, "hipMemcpy(input, H2D)");
        if (hipMemcpy(d_off, offsets + s0, (size_t)ns * 8, hipMemcpyHostToDevice) != hipSuccess) fatal_device(fn, "hipMemcpy(offsets, H2D)");
        if (hipMemcpy(d_gain, gains + s0, (size_t)ns * 8, hipMemcpyHostToDevice) != hipSuccess) fatal_device(fn, "hipMemcpy(gains, H2D)");
        if (fa_int64_to_float64_device(reinterpret_cast<const int64_t*>(d_in), ns, stream_size, d_off, d_gain,
                                       reinterpret_cast<double*>(d_out), nullptr) != FA_ERROR_NONE)
            fatal_device(fn, "int64_to_float64_kernel launch");
        if (hipStreamSynchronize(nullptr) != hipSuccess) fatal_device(fn, "hipStreamSynchronize");
        if (hipMemcpy(output + s0 * stream_size, d_out, nb, hipMemcpyDeviceToHost) != hipSuccess) fatal_device(fn, "hipMemcpy(output, D2H)");
    }
}

}  // extern "C"
