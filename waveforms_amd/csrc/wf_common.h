// wf_common.h — shared host/device helpers for libwfhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>

#include "../../include/wfhip.h"

#define WF_WAVE 64

void wf_set_error(const char *fmt, ...);

#define WF_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            wf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                      \
            return WF_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

#define WF_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            wf_set_error(__VA_ARGS__); \
            return WF_ERR_VALUE;       \
        }                              \
    } while (0)

#define WF_LAUNCH_CHECK() WF_HIP(hipGetLastError())

// Device fault word bits (wf_ctx::d_fault).
enum { WF_FAULT_SCAN_TIMEOUT = 1u };

struct wf_lfsr_tables {
    uint64_t host[64][64];  // host[j][c] = column c of T^(2^j)
    uint64_t *dev = nullptr;  // same, 64*64 words
};

struct wf_ctx {
    int device = 0;
    // chained-scan scratch: [0] ticket counter (as u64), [1..] tile descriptors
    uint64_t *d_scan = nullptr;
    size_t scan_words = 0;
    unsigned *d_fault = nullptr;
    unsigned *h_fault = nullptr;  // pinned
    uint8_t *d_tables = nullptr;  // small table upload area (fsm encode)
    uint64_t *d_fsm_scratch = nullptr;
    size_t fsm_scratch_words = 0;
    int *h_small = nullptr;  // pinned, small D2H results
    int *d_small = nullptr;
    std::map<uint64_t, wf_lfsr_tables *> lfsr;
    hipEvent_t *events = nullptr;  // WF_LINK_EVENT_SLOTS x (WF_LINK_STAGES + 1), created lazily
};

static inline hipStream_t wf_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

int wf_ctx_reserve_scan(wf_ctx *ctx, size_t words);
int wf_ctx_reserve_fsm(wf_ctx *ctx, size_t words);

static inline int wf_grid_for(int64_t work_items, int per_block, int max_blocks)
{
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

#ifdef __HIPCC__
// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ int wf_lane() { return threadIdx.x & (WF_WAVE - 1); }

// Inclusive wave64 prefix sum of doubles.
__device__ __forceinline__ double wf_wave_incl_scan(double v)
{
    const int lane = wf_lane();
#pragma unroll
    for (int d = 1; d < WF_WAVE; d <<= 1) {
        double o = __shfl_up(v, d, WF_WAVE);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ uint64_t wf_wave_xor_reduce(uint64_t v)
{
#pragma unroll
    for (int d = WF_WAVE / 2; d >= 1; d >>= 1) v ^= __shfl_xor(v, d, WF_WAVE);
    return v;
}

__device__ __forceinline__ long long wf_wave_sum_i64(long long v)
{
#pragma unroll
    for (int d = WF_WAVE / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, WF_WAVE);
    return v;
}
#endif
