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
    uint8_t h_tables_cache[2048] = {0};  // what d_tables currently holds (uploads are skipped when unchanged)
    int tables_cached = 0;
    uint64_t *d_fsm_scratch = nullptr;
    size_t fsm_scratch_words = 0;
    int *h_small = nullptr;  // pinned, small D2H results
    int *d_small = nullptr;
    std::map<uint64_t, wf_lfsr_tables *> lfsr;
    double *d_mod_scratch = nullptr;  // fused modulator: constants, tile sums, tile carries
    size_t mod_scratch_words = 0;
    hipEvent_t *events = nullptr;  // WF_LINK_EVENT_SLOTS x (WF_LINK_STAGES + 1), created lazily
};

static inline hipStream_t wf_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

int wf_ctx_reserve_scan(wf_ctx *ctx, size_t words);
int wf_ctx_reserve_fsm(wf_ctx *ctx, size_t words);
int wf_ctx_reserve_mod(wf_ctx *ctx, size_t words);

// Internal (not exported) forms with device-resident carries, used by the streaming link.
int wf_lfsr_generate_dyn(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip,
                         const uint64_t *d_dyn_skip, uint8_t *d_bits, int64_t n, uint64_t *h_state_out, void *stream);
int wf_awgn_mf_bank_dyn(wf_ctx *ctx, const double *d_signal_ri, int64_t nsamp, double rot_re, double rot_im,
                        double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                        const uint64_t *d_dyn_index, const double *d_taps_ri, int nfilt, int ntaps, int64_t first,
                        int step, int64_t ncols, double *d_out_ri, void *stream);
int wf_cpm_modulate_window(wf_ctx *ctx, const int8_t *d_symbols, int64_t sym_origin, int64_t nloc, int64_t nsym_total,
                           const double *d_h, int nh, const double *d_pulse, int ntaps, int sps, double phi0,
                           int64_t tile_lo, int64_t ntiles, double *d_out_ri, int64_t out_origin,
                           const uint64_t *d_q_in, uint64_t *d_q_out, int64_t q_out_tile, void *stream);
int wf_fsm_encode_core(wf_ctx *ctx, const uint8_t *h_next, const int8_t *h_out, int columns, int states, int card,
                       const uint8_t *d_bits, int64_t nbits, int64_t i0, int state0, const int *d_state_in,
                       int8_t *d_symbols, int *h_state_out, int *d_state_at, int64_t at_index, void *stream);

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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global load and store of the wave (s_waitcnt vmcnt(0)), which serialises a
// software pipeline: prefetches issued before the barrier and stores issued after the
// previous one would all have to land first.  Use ONLY where the waves of the workgroup
// exchange data through LDS.
__device__ __forceinline__ void wf_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

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

// sin / cos of 2*pi*t, t in TURNS.  The quadrant reduction is exact (scaling by 4, floor
// and the subtraction are exact in binary floating point), so there is no Payne-Hanek
// path and no cancellation; the residual angle in [0, pi/4] goes through the classic
// minimax kernels (Sun fdlibm __kernel_sin / __kernel_cos coefficient sets, < 1 ulp).
__device__ __forceinline__ void wf_sincos_turns(double t, double *sn, double *cs)
{
    const double y = t * 4.0;
    const double q = floor(y);
    const double f = y - q;                 // [0, 1) quarter turns, exact
    const bool fold = f > 0.5;
    const double g = fold ? 1.0 - f : f;    // [0, 0.5], exact
    const double x = g * 1.57079632679489661923;  // [0, pi/4]
    const double z = x * x;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    ps = fma(x * z, ps, x);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    pc = fma(z * z, pc, fma(z, -0.5, 1.0));
    const double sq = fold ? pc : ps;       // sin / cos inside the quadrant
    const double cq = fold ? ps : pc;
    const int qi = (int)q & 3;
    const double s1 = (qi & 1) ? cq : sq;
    const double c1 = (qi & 1) ? sq : cq;
    *sn = (qi & 2) ? -s1 : s1;
    *cs = ((qi + 1) & 2) ? -c1 : c1;
}

// Natural log of a NORMAL positive double (no zero / subnormal / inf / nan handling:
// callers pass u in [2^-53, 1]).  Exponent split + the fdlibm e_log.c minimax series
// in s = f / (2 + f); the quotient uses v_rcp_f64 + two Newton steps.  < 1 ulp.
__device__ __forceinline__ double wf_log_normal(double x)
{
    const long long ix = __double_as_longlong(x);
    int e = (int)(ix >> 52) - 1023;
    const long long mant = ix & 0x000FFFFFFFFFFFFFll;
    const bool up = mant >= 0x0006A09E667F3BCDll;            // m >= sqrt(2): use m/2
    e += up ? 1 : 0;
    const double m = __longlong_as_double(mant | (up ? 0x3FE0000000000000ll : 0x3FF0000000000000ll));
    const double f = m - 1.0;                                 // [-0.2929, 0.4142]
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = r * fma(-d, r, 2.0);
    r = r * fma(-d, r, 2.0);
    const double s = f * r;
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return fma(dk, 6.93147180369123816490e-01, f - (hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)));
}

// sqrt of a non-negative normal double (0 allowed): v_rsq_f64 seed + Goldschmidt, then
// one residual correction.
__device__ __forceinline__ double wf_sqrt_pos(double a)
{
    const double y = __builtin_amdgcn_rsq(fmax(a, 0x1.0p-1000));
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double dres = fma(-g, g, a);
    return fma(dres, h, g);
}

// ---- Philox4x32-10 + Box-Muller: the device Gaussian source (see wf_awgn.hip) ----
struct wf_philox_out {
    uint32_t x0, x1, x2, x3;
};

__device__ __forceinline__ wf_philox_out wf_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;  // v_mad_u64_u32
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// Box-Muller from two 32-bit words: u1 = (xa + 1) 2^-32 in (0, 1], u2 = xb 2^-32 in [0, 1).
__device__ __forceinline__ void wf_box_muller32(uint32_t xa, uint32_t xb, double sigma, double *re, double *im)
{
    const double u1 = ((double)xa + 1.0) * 0x1.0p-32;
    const double u2 = (double)xb * 0x1.0p-32;
#ifdef WF_ABL_NO_LOG
    const double r = sigma * u1;
#else
    const double r = sigma * wf_sqrt_pos(-2.0 * wf_log_normal(u1));
#endif
    double s, c;
#ifdef WF_ABL_NO_SINCOS
    s = u2; c = 1.0 - u2;
#else
    wf_sincos_turns(u2, &s, &c);
#endif
    *re = r * c;
    *im = r * s;
}

// The two complex Gaussian samples with absolute indices 2*pair and 2*pair + 1: ONE
// Philox4x32-10 block (counter = pair index, stream id; key = seed), words (x0, x1) for the
// even sample, (x2, x3) for the odd one.  g = {re0, im0, re1, im1}.
__device__ __forceinline__ void wf_gaussian_two(uint64_t pair, uint64_t stream_id, uint64_t seed, double sigma,
                                                double g[4])
{
#ifdef WF_ABL_NO_PHILOX   // ablation only: NOT a valid generator
    const wf_philox_out p = {(uint32_t)pair * 2654435761u, (uint32_t)(pair >> 7) ^ (uint32_t)seed,
                             (uint32_t)pair * 40503u, (uint32_t)stream_id ^ (uint32_t)pair};
#else
    const wf_philox_out p = wf_philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)stream_id,
                                             (uint32_t)(stream_id >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
#endif
    wf_box_muller32(p.x0, p.x1, sigma, &g[0], &g[1]);
    wf_box_muller32(p.x2, p.x3, sigma, &g[2], &g[3]);
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
