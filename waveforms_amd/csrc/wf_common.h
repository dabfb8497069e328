// wf_common.h — shared host/device helpers for libwfhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <map>
#include <mutex>
#include <vector>
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
    // linear functionals of a 32768-bit block's base state (wf_lfsr.hip: lfsr_block_functionals): parity of its bits at
    // even / odd positions, its last bit but one and its last bit
    uint64_t f_even = 0, f_odd = 0, f_m2 = 0, f_m1 = 0;
    bool f_ready = false;
};

struct wf_ctx {
    int device = 0;
    int cus = 256;                       // compute units of the device (hipDeviceAttributeMultiprocessorCount at creation)
    int64_t opt[WF_OPT_COUNT] = {0};     // wf_ctx_set_option: 0 = default for every key
    // chained-scan scratch: [0] ticket counter (as u64), [1..] tile descriptors
    uint64_t *d_scan = nullptr;
    size_t scan_words = 0;
    unsigned *d_fault = nullptr;
    unsigned *h_fault = nullptr;  // pinned
    uint8_t *d_tables = nullptr;  // small table upload area (fsm encode)
    uint8_t h_tables_cache[2048] = {0};  // what d_tables currently holds (uploads are skipped when unchanged)
    int tables_cached = 0;
    hipEvent_t tables_event = nullptr;   // recorded behind the last table upload (on tables_stream); cleared once seen complete
    hipStream_t tables_stream = nullptr;
    bool tables_pending = false;
    uint64_t *d_fsm_scratch = nullptr;
    size_t fsm_scratch_words = 0;
    int *h_small = nullptr;  // pinned, small D2H results
    int *d_small = nullptr;
    std::map<uint64_t, wf_lfsr_tables *> lfsr;
    double *d_mod_scratch = nullptr;  // fused modulator: constants, tile sums, tile carries
    size_t mod_scratch_words = 0;
    double *d_vit_edge = nullptr;     // chunk-parallel detectors: per-chunk proof records + repair lists of the last launch
    size_t vit_edge_words = 0;
    unsigned long long *d_vit_unmerged = nullptr;   // [0] chunks left unproven, [1] chunk repairs run, [2] repairs that handed on to the next chunk, [3] spare
    hipEvent_t *events = nullptr;  // WF_LINK_EVENT_SLOTS x (WF_LINK_STAGES + 1), created lazily
    // wf_link_run with fuse bit 5: the detector and the error count of a block run on this side stream, beside the front
    // end of the NEXT block (two sets of intermediates in the workspace, used alternately).  pipe_done[s]: the back end
    // of the last block that used set s; wf_link_join / wf_ctx_check make the caller's stream wait for both.
    void *pipe_stream = nullptr;
    void *pipe_stream2 = nullptr;        // the CPM links alternate consecutive blocks' back ends between two side streams (wf_cpm_link_run)
    // the pipelined SOQPSK link (wf_link_run): the NEXT block's prologue (PRBS + precoder, tile sums, tile scan) on pipe_pro while
    // this block's front end runs on pipe_main, whose CU mask keeps a few compute units free for those small kernels
    void *pipe_pro = nullptr, *pipe_main = nullptr;
    hipEvent_t pipe_call = nullptr, pipe_pro_done = nullptr;
    hipEvent_t pipe_front = nullptr, pipe_done[2] = {nullptr, nullptr};
    bool pipe_done_valid[2] = {false, false};
    int pipe_set = 0;
    // caller promises about small operand tables that the library has verified on the host (wf_promise_verified): key =
    // hash of (kind, device pointers, sizes), value = uses since the last verification
    std::map<uint64_t, uint32_t> promises;
    std::mutex promises_lock;
    double *h_iter = nullptr;      // per-symbol detector call: pinned, device-mapped staging (6 in + 2 x 64 out)
    double *d_iter = nullptr;      // ... the device's address of the same memory
    void *h_mailbox = nullptr;     // per-symbol detector call through the persistent iteration server (wf_viterbi.hip): pinned mailbox,
    void *d_mailbox = nullptr;     // ... its device address,
    void *iter_stream = nullptr;   // ... the server's own (non-blocking) stream,
    const void *iter_last_state = nullptr;   // ... and the detector state the last request was for
    std::atomic_flag iter_lock = ATOMIC_FLAG_INIT;   // the mailbox holds ONE request: callers of the per-symbol entry point take turns
    bool iter_closing = false;     // set (under iter_lock) when the context is retired: no new server is started on it
};

static inline hipStream_t wf_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

int wf_ctx_reserve_scan(wf_ctx *ctx, size_t words);
int wf_ctx_reserve_fsm(wf_ctx *ctx, size_t words);
int wf_ctx_reserve_mod(wf_ctx *ctx, size_t words);
int wf_iter_server_stop(wf_ctx *ctx);
int wf_ctx_reserve_vit(wf_ctx *ctx, size_t words);

struct lfsr_emit_args;
// Internal (not exported) forms with device-resident carries, used by the streaming link.
int wf_lfsr_generate_dyn(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip,
                         const uint64_t *d_dyn_skip, uint8_t *d_bits, int64_t n, uint64_t *h_state_out, void *stream,
                         const struct lfsr_emit_args *emit = nullptr);   // emit: the link's precoder rides in the same launch (see lfsr_kernel)
// PRBS + the SOQPSK 4-state precoder of the link in ONE launch (wf_lfsr.hip); 1 = not this trellis / too long: use the generic calls
int wf_lfsr_generate_map(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip, uint8_t *d_bits, int64_t nbits,
                         int map_kind, int8_t *d_symbols, void *stream);   // PRBS + memoryless CPM mapper in one launch (1: not this mapper)
int wf_soqpsk_prbs_encode(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip, const uint8_t *h_next,
                          const int8_t *h_out, uint8_t *d_bits, int64_t n, int8_t *d_symbols, void *stream, void *mid_event = nullptr);
int wf_awgn_mf_bank_dyn(wf_ctx *ctx, const double *d_signal_ri, int64_t nsamp, double rot_re, double rot_im,
                        double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                        const uint64_t *d_dyn_index, const double *d_taps_ri, int nfilt, int ntaps, int64_t first,
                        int step, int64_t ncols, double *d_out_ri, void *stream, int pack_par0 = -1);
// wf_viterbi4_detect over detector-packed rows (4 doubles per call, see mf_bank_kernel PACK)
int wf_viterbi4_detect_packed(wf_ctx *ctx, const double *d_rows4, int64_t ncalls, int differential, int warmup,
                              uint8_t *d_bits, int8_t *d_syms, double *d_state, void *stream);
int wf_cpm_modulate_window(wf_ctx *ctx, const int8_t *d_symbols, int64_t sym_origin, int64_t nloc, int64_t nsym_total,
                           const double *d_h, int nh, const double *d_pulse, int ntaps, int sps, double phi0,
                           int64_t tile_lo, int64_t ntiles, double *d_out_ri, int64_t out_origin,
                           const uint64_t *d_q_in, uint64_t *d_q_out, int64_t q_out_tile, void *stream);
// Per-launch options of the one-kernel front end, passed BY ARGUMENT (never parked on the shared context: links on one
// context may run on several host threads).  pam_factor: the long (PAM) bank factored into two real filters + a 3 x 2
// complex combination (wf_link_config.d_mf_factor, verified against d_mf_taps by the link) or NULL; cpm_paired: the CPM
// templates pair off as conjugates (wf_cpm_link_config.fuse bit 6, verified by the link); runs_hint: runs of tiles per
// resident slot for the CPM forms (0: default; the pipelined CPM link asks for finer runs).
struct wf_mcb_opts {
    const double *pam_factor = nullptr;
    bool cpm_paired = false;
    int runs_hint = 0;
    int scratch_slot = -1;      // 0 / 1: which of two sets of carries in the context's scratch (stage 1 of one block beside stage 2 of another); -1: the one set
};
int wf_mod_chan_bank_packed(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h, int nh,
                            const double *d_pulse, int ntaps, int sps, double phi0, const double *d_mf_taps, double rot_re,
                            double rot_im, double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                            int64_t first, int64_t ncols, int pack_par0, double *d_rows4, void *stream, int mf_ntaps = 0,
                            const wf_mcb_opts *opts = nullptr, int stage = 3);   // stage 1: the carry kernels only, 2: the main kernel on carries a stage-1 call left (wf_mod_chan_bank_window)
int wf_mod_chan_bank_window(wf_ctx *ctx, const int8_t *d_symbols, int64_t sym_origin, int64_t nloc, int64_t nsym_total,
                            const double *d_h, int nh, const double *d_pulse, int ntaps, int sps, double phi0,
                            int64_t tile_lo, int64_t ntiles, const uint64_t *d_q_in, uint64_t *d_q_out, int64_t q_out_tile,
                            const double *d_mf_taps, double rot_re, double rot_im, double sigma, uint64_t seed,
                            uint64_t stream_id, uint64_t first_index, const uint64_t *d_dyn_index, int64_t first,
                            int64_t k_lo, int64_t ncols, int pack_par0, double *d_rows4, void *stream, int cpm_nf = 0, int cpm_nh = 1,
                            int stage = 3, int mf_ntaps = 0, const wf_mcb_opts *opts = nullptr);
int wf_mod_chan_bank_applies(int64_t nsym, int nh, int ntaps, int sps, int mf_ntaps, int64_t first);
int wf_link_join_internal(wf_ctx *ctx, void *stream);   // the caller's stream waits for what fuse bit 5 left on the side stream
int wf_mod_chan_cpm_rows_applies(int64_t nsym, int nh, int ntaps, int sps, int nfilt, int ntm, int64_t start0);
int wf_mod_chan_cpm_rows(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h, int nh, const double *d_pulse,
                         int ntaps, int sps, double phi0, const double *d_templates, int nfilt, int ntm, int64_t start0,
                         double rot_re, double rot_im, double sigma, uint64_t seed, uint64_t stream_id, int64_t ncalls,
                         double *d_rows, void *stream, const wf_mcb_opts *opts = nullptr);
// Modulator + channel with the NOISY SAMPLES stored (the front end of the CPM link whose detector runs the matched filters itself,
// wf_cpm_link_config.fuse bit 7): 1 = configuration outside the kernel.
int wf_mod_chan_samples_applies(int64_t nsym, int nh, int ntaps, int sps);
int wf_mod_chan_samples(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h, int nh, const double *d_pulse, int ntaps,
                        int sps, double phi0, double rot_re, double rot_im, double sigma, uint64_t seed, uint64_t stream_id,
                        uint64_t first_index, double *d_out_ri, void *stream);
int wf_mod_chan_samples_window(wf_ctx *ctx, const int8_t *d_symbols, int64_t sym_origin, int64_t nloc, int64_t nsym_total, const double *d_h, int nh,
                               const double *d_pulse, int ntaps, int sps, double phi0, int64_t tile_lo, int64_t ntiles, const uint64_t *d_q_in,
                               uint64_t *d_q_out, int64_t q_out_tile, double rot_re, double rot_im, double sigma, uint64_t seed, uint64_t stream_id,
                               uint64_t first_index, double *d_out_ri, int64_t out_origin, int64_t out_hi, void *stream);
// A promise a caller makes about small device operands (<= 64 KB), checked on the HOST the first time this content is
// seen on this context: `check(host copy, nbytes)` decides; the verdict is cached under a hash of (kind, device ADDRESSES, sizes)
// — wf_ctx_forget_promises drops the cache when tables are freed or rewritten (the links call it on creation).
// Synchronises `stream` on a cache miss only.  Returns WF_OK, or WF_ERR_VALUE with `what` in the error text.
int wf_promise_verified(wf_ctx *ctx, int kind, const void *const *d_ptrs, const size_t *nbytes, int nptrs, void *stream,
                        bool (*check)(const unsigned char *const *host, const size_t *nbytes, const void *arg), const void *arg, const char *what);
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

// 16 B accesses with the non-temporal hint, for the 1.28 GB of baseband samples the modulator
// writes once and the next kernel reads once: the lines do not linger in L2 as dirty data.
// Measured inside the link: modulator 0.297 -> 0.284 ms and the channel kernel that reads those
// samples 0.408 -> 0.391 ms.  (The same hint on the bank's row stores, on its sample loads and on
// the detector's row loads changed nothing.)
typedef double wf_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wf_store16_nt(double2 *p, double2 v)
{
    wf_v2d t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<wf_v2d *>(p));
}
__device__ __forceinline__ double2 wf_load16_nt(const double2 *p)
{
    const wf_v2d t = __builtin_nontemporal_load(reinterpret_cast<const wf_v2d *>(p));
    return make_double2(t.x, t.y);
}

// A double moved across lanes by DPP (two v_mov_b32_dpp): lanes without a source, or in rows
// outside ROW_MASK, receive 0.0.  CTRL: 0x110 + n = row_shr:n, 0x142 = row_bcast:15,
// 0x143 = row_bcast:31, 0x138 = wave_shr:1.
// With every row enabled, bound_ctrl lets the hardware supply the zeros (no register has to be
// cleared first); with a row mask the untouched rows keep the zero passed as `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double wf_dpp_f64(double v)
{
    constexpr bool BC = ROW_MASK == 0xf;
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, BC);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, BC);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// Inclusive wave64 prefix sum of doubles, entirely in the VALU: Hillis-Steele inside each row of
// 16 lanes (row_shr 1, 2, 4, 8), then the row totals are broadcast into the following rows
// (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  18 instructions and no LDS
// round trips — as __shfl_up (ds_bpermute_b32) the same scan is a chain of 6 dependent LDS
// accesses, which was the latency that bounded the modulator's row loop.
__device__ __forceinline__ double wf_wave_incl_scan(double v)
{
    v += wf_dpp_f64<0x111, 0xf>(v);
    v += wf_dpp_f64<0x112, 0xf>(v);
    v += wf_dpp_f64<0x114, 0xf>(v);
    v += wf_dpp_f64<0x118, 0xf>(v);
    v += wf_dpp_f64<0x142, 0xa>(v);
    v += wf_dpp_f64<0x143, 0xc>(v);
    return v;
}

// The value of the lane below (0.0 in lane 0).
__device__ __forceinline__ double wf_wave_shr1(double v) { return wf_dpp_f64<0x138, 0xf>(v); }

// sin / cos of 2*pi*t: table-driven.  The turn is cut into 128 sectors; {cos, sin} at the sector
// centres come from a 2 KB table (generated in 80-bit arithmetic, rounded once), the residual
// angle |r| <= pi/128 goes through two short Taylor kernels (first terms left out: r^7/5040 < 1.1e-15 in the sine,
// r^8/8! < 4e-18 in the cosine) and one complex rotation.  The sector split is exact (scaling by 128, floor
// and the subtractions are exact in binary floating point; for a 32-bit uniform it is pure
// integer work), so there is no Payne-Hanek path, no quadrant selects and no cancellation.
// Absolute error < 1.5e-15; 13 fp64 operations (the r^7 term of the sine, kept through round 4, was the 14th: an
// issue-bound kernel pays for every one of them, profiles/r05_ab_numerics.log) against 18 + 16 selects for the octant form.
static __device__ __constant__ double2 kWfCisTab[128] = {
    {0x1.ffd886084cd0dp-1, 0x1.92155f7a3667ep-6}, {0x1.fe9cdad01883ap-1, 0x1.2d52092ce19f6p-4},
    {0x1.fc26470e19fd3p-1, 0x1.f564e56a9730ep-4}, {0x1.f8764fa714ba9p-1, 0x1.5e214448b3fc6p-3},
    {0x1.f38f3ac64e589p-1, 0x1.c0b826a7e4f63p-3}, {0x1.ed740e7684963p-1, 0x1.111d262b1f677p-2},
    {0x1.e6288ec48e112p-1, 0x1.4135c94176601p-2}, {0x1.ddb13b6ccc23cp-1, 0x1.7088530fa459fp-2},
    {0x1.d4134d14dc93ap-1, 0x1.9ef7943a8ed8ap-2}, {0x1.c954b213411f5p-1, 0x1.cc66e9931c45ep-2},
    {0x1.bd7c0ac6f952ap-1, 0x1.f8ba4dbf89abap-2}, {0x1.b090a58150200p-1, 0x1.11eb3541b4b23p-1},
    {0x1.a29a7a0462782p-1, 0x1.26d054cdd12dfp-1}, {0x1.93a22499263fbp-1, 0x1.3affa292050b9p-1},
    {0x1.83b0e0bff976ep-1, 0x1.4e6cabbe3e5e9p-1}, {0x1.72d0837efff96p-1, 0x1.610b7551d2cdfp-1},
    {0x1.610b7551d2cdfp-1, 0x1.72d0837efff96p-1}, {0x1.4e6cabbe3e5e9p-1, 0x1.83b0e0bff976ep-1},
    {0x1.3affa292050b9p-1, 0x1.93a22499263fbp-1}, {0x1.26d054cdd12dfp-1, 0x1.a29a7a0462782p-1},
    {0x1.11eb3541b4b23p-1, 0x1.b090a58150200p-1}, {0x1.f8ba4dbf89abap-2, 0x1.bd7c0ac6f952ap-1},
    {0x1.cc66e9931c45ep-2, 0x1.c954b213411f5p-1}, {0x1.9ef7943a8ed8ap-2, 0x1.d4134d14dc93ap-1},
    {0x1.7088530fa459fp-2, 0x1.ddb13b6ccc23cp-1}, {0x1.4135c94176601p-2, 0x1.e6288ec48e112p-1},
    {0x1.111d262b1f677p-2, 0x1.ed740e7684963p-1}, {0x1.c0b826a7e4f63p-3, 0x1.f38f3ac64e589p-1},
    {0x1.5e214448b3fc6p-3, 0x1.f8764fa714ba9p-1}, {0x1.f564e56a9730ep-4, 0x1.fc26470e19fd3p-1},
    {0x1.2d52092ce19f6p-4, 0x1.fe9cdad01883ap-1}, {0x1.92155f7a3667ep-6, 0x1.ffd886084cd0dp-1},
    {-0x1.92155f7a3667ep-6, 0x1.ffd886084cd0dp-1}, {-0x1.2d52092ce19f6p-4, 0x1.fe9cdad01883ap-1},
    {-0x1.f564e56a9730ep-4, 0x1.fc26470e19fd3p-1}, {-0x1.5e214448b3fc6p-3, 0x1.f8764fa714ba9p-1},
    {-0x1.c0b826a7e4f63p-3, 0x1.f38f3ac64e589p-1}, {-0x1.111d262b1f677p-2, 0x1.ed740e7684963p-1},
    {-0x1.4135c94176601p-2, 0x1.e6288ec48e112p-1}, {-0x1.7088530fa459fp-2, 0x1.ddb13b6ccc23cp-1},
    {-0x1.9ef7943a8ed8ap-2, 0x1.d4134d14dc93ap-1}, {-0x1.cc66e9931c45ep-2, 0x1.c954b213411f5p-1},
    {-0x1.f8ba4dbf89abap-2, 0x1.bd7c0ac6f952ap-1}, {-0x1.11eb3541b4b23p-1, 0x1.b090a58150200p-1},
    {-0x1.26d054cdd12dfp-1, 0x1.a29a7a0462782p-1}, {-0x1.3affa292050b9p-1, 0x1.93a22499263fbp-1},
    {-0x1.4e6cabbe3e5e9p-1, 0x1.83b0e0bff976ep-1}, {-0x1.610b7551d2cdfp-1, 0x1.72d0837efff96p-1},
    {-0x1.72d0837efff96p-1, 0x1.610b7551d2cdfp-1}, {-0x1.83b0e0bff976ep-1, 0x1.4e6cabbe3e5e9p-1},
    {-0x1.93a22499263fbp-1, 0x1.3affa292050b9p-1}, {-0x1.a29a7a0462782p-1, 0x1.26d054cdd12dfp-1},
    {-0x1.b090a58150200p-1, 0x1.11eb3541b4b23p-1}, {-0x1.bd7c0ac6f952ap-1, 0x1.f8ba4dbf89abap-2},
    {-0x1.c954b213411f5p-1, 0x1.cc66e9931c45ep-2}, {-0x1.d4134d14dc93ap-1, 0x1.9ef7943a8ed8ap-2},
    {-0x1.ddb13b6ccc23cp-1, 0x1.7088530fa459fp-2}, {-0x1.e6288ec48e112p-1, 0x1.4135c94176601p-2},
    {-0x1.ed740e7684963p-1, 0x1.111d262b1f677p-2}, {-0x1.f38f3ac64e589p-1, 0x1.c0b826a7e4f63p-3},
    {-0x1.f8764fa714ba9p-1, 0x1.5e214448b3fc6p-3}, {-0x1.fc26470e19fd3p-1, 0x1.f564e56a9730ep-4},
    {-0x1.fe9cdad01883ap-1, 0x1.2d52092ce19f6p-4}, {-0x1.ffd886084cd0dp-1, 0x1.92155f7a3667ep-6},
    {-0x1.ffd886084cd0dp-1, -0x1.92155f7a3667ep-6}, {-0x1.fe9cdad01883ap-1, -0x1.2d52092ce19f6p-4},
    {-0x1.fc26470e19fd3p-1, -0x1.f564e56a9730ep-4}, {-0x1.f8764fa714ba9p-1, -0x1.5e214448b3fc6p-3},
    {-0x1.f38f3ac64e589p-1, -0x1.c0b826a7e4f63p-3}, {-0x1.ed740e7684963p-1, -0x1.111d262b1f677p-2},
    {-0x1.e6288ec48e112p-1, -0x1.4135c94176601p-2}, {-0x1.ddb13b6ccc23cp-1, -0x1.7088530fa459fp-2},
    {-0x1.d4134d14dc93ap-1, -0x1.9ef7943a8ed8ap-2}, {-0x1.c954b213411f5p-1, -0x1.cc66e9931c45ep-2},
    {-0x1.bd7c0ac6f952ap-1, -0x1.f8ba4dbf89abap-2}, {-0x1.b090a58150200p-1, -0x1.11eb3541b4b23p-1},
    {-0x1.a29a7a0462782p-1, -0x1.26d054cdd12dfp-1}, {-0x1.93a22499263fbp-1, -0x1.3affa292050b9p-1},
    {-0x1.83b0e0bff976ep-1, -0x1.4e6cabbe3e5e9p-1}, {-0x1.72d0837efff96p-1, -0x1.610b7551d2cdfp-1},
    {-0x1.610b7551d2cdfp-1, -0x1.72d0837efff96p-1}, {-0x1.4e6cabbe3e5e9p-1, -0x1.83b0e0bff976ep-1},
    {-0x1.3affa292050b9p-1, -0x1.93a22499263fbp-1}, {-0x1.26d054cdd12dfp-1, -0x1.a29a7a0462782p-1},
    {-0x1.11eb3541b4b23p-1, -0x1.b090a58150200p-1}, {-0x1.f8ba4dbf89abap-2, -0x1.bd7c0ac6f952ap-1},
    {-0x1.cc66e9931c45ep-2, -0x1.c954b213411f5p-1}, {-0x1.9ef7943a8ed8ap-2, -0x1.d4134d14dc93ap-1},
    {-0x1.7088530fa459fp-2, -0x1.ddb13b6ccc23cp-1}, {-0x1.4135c94176601p-2, -0x1.e6288ec48e112p-1},
    {-0x1.111d262b1f677p-2, -0x1.ed740e7684963p-1}, {-0x1.c0b826a7e4f63p-3, -0x1.f38f3ac64e589p-1},
    {-0x1.5e214448b3fc6p-3, -0x1.f8764fa714ba9p-1}, {-0x1.f564e56a9730ep-4, -0x1.fc26470e19fd3p-1},
    {-0x1.2d52092ce19f6p-4, -0x1.fe9cdad01883ap-1}, {-0x1.92155f7a3667ep-6, -0x1.ffd886084cd0dp-1},
    {0x1.92155f7a3667ep-6, -0x1.ffd886084cd0dp-1}, {0x1.2d52092ce19f6p-4, -0x1.fe9cdad01883ap-1},
    {0x1.f564e56a9730ep-4, -0x1.fc26470e19fd3p-1}, {0x1.5e214448b3fc6p-3, -0x1.f8764fa714ba9p-1},
    {0x1.c0b826a7e4f63p-3, -0x1.f38f3ac64e589p-1}, {0x1.111d262b1f677p-2, -0x1.ed740e7684963p-1},
    {0x1.4135c94176601p-2, -0x1.e6288ec48e112p-1}, {0x1.7088530fa459fp-2, -0x1.ddb13b6ccc23cp-1},
    {0x1.9ef7943a8ed8ap-2, -0x1.d4134d14dc93ap-1}, {0x1.cc66e9931c45ep-2, -0x1.c954b213411f5p-1},
    {0x1.f8ba4dbf89abap-2, -0x1.bd7c0ac6f952ap-1}, {0x1.11eb3541b4b23p-1, -0x1.b090a58150200p-1},
    {0x1.26d054cdd12dfp-1, -0x1.a29a7a0462782p-1}, {0x1.3affa292050b9p-1, -0x1.93a22499263fbp-1},
    {0x1.4e6cabbe3e5e9p-1, -0x1.83b0e0bff976ep-1}, {0x1.610b7551d2cdfp-1, -0x1.72d0837efff96p-1},
    {0x1.72d0837efff96p-1, -0x1.610b7551d2cdfp-1}, {0x1.83b0e0bff976ep-1, -0x1.4e6cabbe3e5e9p-1},
    {0x1.93a22499263fbp-1, -0x1.3affa292050b9p-1}, {0x1.a29a7a0462782p-1, -0x1.26d054cdd12dfp-1},
    {0x1.b090a58150200p-1, -0x1.11eb3541b4b23p-1}, {0x1.bd7c0ac6f952ap-1, -0x1.f8ba4dbf89abap-2},
    {0x1.c954b213411f5p-1, -0x1.cc66e9931c45ep-2}, {0x1.d4134d14dc93ap-1, -0x1.9ef7943a8ed8ap-2},
    {0x1.ddb13b6ccc23cp-1, -0x1.7088530fa459fp-2}, {0x1.e6288ec48e112p-1, -0x1.4135c94176601p-2},
    {0x1.ed740e7684963p-1, -0x1.111d262b1f677p-2}, {0x1.f38f3ac64e589p-1, -0x1.c0b826a7e4f63p-3},
    {0x1.f8764fa714ba9p-1, -0x1.5e214448b3fc6p-3}, {0x1.fc26470e19fd3p-1, -0x1.f564e56a9730ep-4},
    {0x1.fe9cdad01883ap-1, -0x1.2d52092ce19f6p-4}, {0x1.ffd886084cd0dp-1, -0x1.92155f7a3667ep-6},
};

__device__ __forceinline__ void wf_cis_sector(double2 cs0, double r, double *sn, double *cs)
{
    const double z = r * r;
    const double ps = fma(z, 1.0 / 120.0, -1.0 / 6.0);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.0 / 720.0, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    const double c = fma(z, pc, 1.0);
    *cs = fma(-cs0.y, s, cs0.x * c);
    *sn = fma(cs0.x, s, cs0.y * c);
}

// `tab` is kWfCisTab or a copy of it in LDS (wf_stage_cis_table): on gfx9-family hardware loads
// and stores share the in-order vmcnt counter, so a table load from global memory inside a loop
// that also stores makes every iteration wait for the previous iteration's stores to land.
__device__ __forceinline__ void wf_sincos_sectors(const double2 *tab, double y, double *sn, double *cs)   // angle = y / 128 turns
{
    // y - floor(y) is v_fract_f64 (exact), and the centring rides in the scaling's fma
    const double r = fma(__builtin_amdgcn_fract(y), 6.28318530717958647692 / 128.0, -0.5 * (6.28318530717958647692 / 128.0));
    wf_cis_sector(tab[(int)floor(y) & 127], r, sn, cs);
}
// ... for y >= 0 (the modulator's phase in 1/128 turns with a non-negative offset): truncation IS the floor, one operation less
__device__ __forceinline__ void wf_sincos_sectors_pos(const double2 *tab, double y, double *sn, double *cs)
{
    const double r = fma(__builtin_amdgcn_fract(y), 6.28318530717958647692 / 128.0, -0.5 * (6.28318530717958647692 / 128.0));
    wf_cis_sector(tab[(int)y & 127], r, sn, cs);
}

// Copy the sector table into LDS (128 x 16 B); the caller synchronises before the first use.
__device__ __forceinline__ void wf_stage_cis_table(double2 *lds_tab, int t, int nthreads)
{
    for (int k = t; k < 128; k += nthreads) lds_tab[k] = kWfCisTab[k];
}


// ln((xa + 1) * 2^-32) for a 32-bit word: table-driven.  m in [1, 2) from the exponent split of
// the exact double xa + 1; 128-entry table of {1/c_i (rounded), ln(1/that)} at the bucket
// centres c_i = 1 + (i + 0.5)/128 (generated with 50-digit arithmetic), r = m/c_i - 1 by one
// fma (|r| <= 2^-8), log1p(r) by a degree-6 series (next term < 2e-18).  About half the fp64
// operations of the division-based series, no rcp.  < 1 ulp.
static __device__ __constant__ double2 kWfLogTab[128] = {
    {0x1.fe01fe01fe020p-1, 0x1.ff00aa2b10ba0p-9}, {0x1.fa11caa01fa12p-1, 0x1.7dc475f810a69p-7},
    {0x1.f6310aca0dbb5p-1, 0x1.3cea44346a584p-6}, {0x1.f25f644230ab5p-1, 0x1.b9fc027af919ap-6},
    {0x1.ee9c7f8458e02p-1, 0x1.1b0d98923d97fp-5}, {0x1.eae807aba01ebp-1, 0x1.58a5bafc8e4d3p-5},
    {0x1.e741aa59750e4p-1, 0x1.95c830ec8e3f2p-5}, {0x1.e3a9179dc1a73p-1, 0x1.d276b8adb0b56p-5},
    {0x1.e01e01e01e01ep-1, 0x1.075983598e471p-4}, {0x1.dca01dca01dcap-1, 0x1.253f62f0a1417p-4},
    {0x1.d92f2231e7f8ap-1, 0x1.42edcbea646eep-4}, {0x1.d5cac807572b2p-1, 0x1.60658a93750c4p-4},
    {0x1.d272ca3fc5b1ap-1, 0x1.7da766d7b12d0p-4}, {0x1.cf26e5c44bfc6p-1, 0x1.9ab42462033aep-4},
    {0x1.cbe6d9601cbe7p-1, 0x1.b78c82bb0eda0p-4}, {0x1.c8b265afb8a42p-1, 0x1.d4313d66cb35dp-4},
    {0x1.c5894d10d4986p-1, 0x1.f0a30c01162a4p-4}, {0x1.c26b5392ea01cp-1, 0x1.0671512ca596fp-3},
    {0x1.bf583ee868d8bp-1, 0x1.14785846742acp-3}, {0x1.bc4fd65883e7bp-1, 0x1.2266f190a5acdp-3},
    {0x1.b951e2b18ff23p-1, 0x1.303d718e47fd5p-3}, {0x1.b65e2e3beee05p-1, 0x1.3dfc2b0ecc62ap-3},
    {0x1.b37484ad806cep-1, 0x1.4ba36f39a55e5p-3}, {0x1.b094b31d922a4p-1, 0x1.59338d9982085p-3},
    {0x1.adbe87f94905ep-1, 0x1.66acd4272ad51p-3}, {0x1.aaf1d2f87ebfdp-1, 0x1.740f8f54037a3p-3},
    {0x1.a82e65130e159p-1, 0x1.815c0a14357e9p-3}, {0x1.a574107688a4ap-1, 0x1.8e928de886d41p-3},
    {0x1.a2c2a87c51ca0p-1, 0x1.9bb362e7dfb85p-3}, {0x1.a01a01a01a01ap-1, 0x1.a8becfc882f19p-3},
    {0x1.9d79f176b682dp-1, 0x1.b5b519e8fb5a6p-3}, {0x1.9ae24ea5510dap-1, 0x1.c2968558c18c2p-3},
    {0x1.9852f0d8ec0ffp-1, 0x1.cf6354e09c5ddp-3}, {0x1.95cbb0be377aep-1, 0x1.dc1bca0abec7bp-3},
    {0x1.934c67f9b2ce6p-1, 0x1.e8c0252aa5a60p-3}, {0x1.90d4f120190d5p-1, 0x1.f550a564b7b37p-3},
    {0x1.8e6527af1373fp-1, 0x1.00e6c45ad501dp-2}, {0x1.8bfce8062ff3ap-1, 0x1.071b85fcd590dp-2},
    {0x1.899c0f601899cp-1, 0x1.0d46b579ab74bp-2}, {0x1.87427bcc092b9p-1, 0x1.136870293a8b0p-2},
    {0x1.84f00c2780614p-1, 0x1.1980d2dd4236fp-2}, {0x1.82a4a0182a4a0p-1, 0x1.1f8ff9e48a2f3p-2},
    {0x1.8060180601806p-1, 0x1.2596010df763ap-2}, {0x1.7e225515a4f1dp-1, 0x1.2b9303ab89d25p-2},
    {0x1.7beb3922e017cp-1, 0x1.31871c9544185p-2}, {0x1.79baa6bb6398bp-1, 0x1.3772662bfd85cp-2},
    {0x1.77908119ac60dp-1, 0x1.3d54fa5c1f710p-2}, {0x1.756cac201756dp-1, 0x1.432ef2a04e813p-2},
    {0x1.734f0c541fe8dp-1, 0x1.49006804009d0p-2}, {0x1.713786d9c7c09p-1, 0x1.4ec9732600269p-2},
    {0x1.6f26016f26017p-1, 0x1.548a2c3add263p-2}, {0x1.6d1a62681c861p-1, 0x1.5a42ab0f4cfe2p-2},
    {0x1.6b1490aa31a3dp-1, 0x1.5ff3070a793d4p-2}, {0x1.691473a88d0c0p-1, 0x1.659b57303e1f2p-2},
    {0x1.6719f3601671ap-1, 0x1.6b3bb2235943dp-2}, {0x1.6524f853b4aa3p-1, 0x1.70d42e2789236p-2},
    {0x1.63356b88ac0dep-1, 0x1.7664e1239dbcfp-2}, {0x1.614b36831ae94p-1, 0x1.7bede0a37afbfp-2},
    {0x1.5f66434292dfcp-1, 0x1.816f41da0d495p-2}, {0x1.5d867c3ece2a5p-1, 0x1.86e919a330ba1p-2},
    {0x1.5babcc647fa91p-1, 0x1.8c5b7c858b48bp-2}, {0x1.59d61f123ccaap-1, 0x1.91c67eb45a83ep-2},
    {0x1.5805601580560p-1, 0x1.972a341135159p-2}, {0x1.56397ba7c52e2p-1, 0x1.9c86b02dc0862p-2},
    {0x1.54725e6bb82fep-1, 0x1.a1dc064d5b995p-2}, {0x1.52aff56a8054bp-1, 0x1.a72a4966bd9e9p-2},
    {0x1.50f22e111c4c5p-1, 0x1.ac718c258b0e5p-2}, {0x1.4f38f62dd4c9bp-1, 0x1.b1b1e0ebdfc5ap-2},
    {0x1.4d843bedc2c4cp-1, 0x1.b6eb59d3cf35cp-2}, {0x1.4bd3edda68fe1p-1, 0x1.bc1e08b0dad0ap-2},
    {0x1.4a27fad76014ap-1, 0x1.c149ff115f027p-2}, {0x1.4880522014880p-1, 0x1.c66f4e3ff6ff9p-2},
    {0x1.46dce34596066p-1, 0x1.cb8e0744d7acap-2}, {0x1.453d9e2c776cap-1, 0x1.d0a63ae721e64p-2},
    {0x1.43a2730abee4dp-1, 0x1.d5b7f9ae2c684p-2}, {0x1.420b5265e5951p-1, 0x1.dac353e2c5955p-2},
    {0x1.40782d10e6566p-1, 0x1.dfc859906d5b5p-2}, {0x1.3ee8f42a5af07p-1, 0x1.e4c71a8687704p-2},
    {0x1.3d5d991aa75c6p-1, 0x1.e9bfa659861f5p-2}, {0x1.3bd60d9232955p-1, 0x1.eeb20c640ddf3p-2},
    {0x1.3a524387ac822p-1, 0x1.f39e5bc811e5dp-2}, {0x1.38d22d366088ep-1, 0x1.f884a36fe9ec1p-2},
    {0x1.3755bd1c945eep-1, 0x1.fd64f20f61571p-2}, {0x1.35dce5f9f2af8p-1, 0x1.011fab125ff8ap-1},
    {0x1.34679ace01346p-1, 0x1.0389eefce633cp-1}, {0x1.32f5ced6a1dfap-1, 0x1.05f14bd26459cp-1},
    {0x1.3187758e9ebb6p-1, 0x1.0855c884b450ep-1}, {0x1.301c82ac40260p-1, 0x1.0ab76bece14d2p-1},
    {0x1.2eb4ea1fed14bp-1, 0x1.0d163ccb9d6b8p-1}, {0x1.2d50a012d50a0p-1, 0x1.0f7241c9b497dp-1},
    {0x1.2bef98e5a3711p-1, 0x1.11cb81787ccf8p-1}, {0x1.2a91c92f3c105p-1, 0x1.1422025243d45p-1},
    {0x1.293725bb804a5p-1, 0x1.1675cababa60ep-1}, {0x1.27dfa38a1ce4dp-1, 0x1.18c6e0ff5cf07p-1},
    {0x1.268b37cd60127p-1, 0x1.1b154b57da29ep-1}, {0x1.2539d7e9177b2p-1, 0x1.1d610fe677003p-1},
    {0x1.23eb79717605bp-1, 0x1.1faa34b87094cp-1}, {0x1.22a0122a0122ap-1, 0x1.21f0bfc65beecp-1},
    {0x1.21579804855e6p-1, 0x1.2434b6f483934p-1}, {0x1.2012012012012p-1, 0x1.26762013430e0p-1},
    {0x1.1ecf43c7fb84cp-1, 0x1.28b500df60783p-1}, {0x1.1d8f5672e4abdp-1, 0x1.2af15f02640acp-1},
    {0x1.1c522fc1ce059p-1, 0x1.2d2b4012edc9dp-1}, {0x1.1b17c67f2bae3p-1, 0x1.2f62a99509546p-1},
    {0x1.19e0119e0119ep-1, 0x1.3197a0fa7fe6ap-1}, {0x1.18ab083902bdbp-1, 0x1.33ca2ba328994p-1},
    {0x1.1778a191bd684p-1, 0x1.35fa4edd36ea0p-1}, {0x1.1648d50fc3201p-1, 0x1.38280fe58797fp-1},
    {0x1.151b9a3fdd5c9p-1, 0x1.3a5373e7ebdf9p-1}, {0x1.13f0e8d344724p-1, 0x1.3c7c7fff73206p-1},
    {0x1.12c8b89edc0acp-1, 0x1.3ea33936b2f5bp-1}, {0x1.11a3019a74826p-1, 0x1.40c7a4880dceap-1},
    {0x1.107fbbe011080p-1, 0x1.42e9c6ddf80bfp-1}, {0x1.0f5edfab325a2p-1, 0x1.4509a5133bb0ap-1},
    {0x1.0e40655826011p-1, 0x1.472743f33aaadp-1}, {0x1.0d24456359e3ap-1, 0x1.4942a83a2fc07p-1},
    {0x1.0c0a7868b4171p-1, 0x1.4b5bd6956e273p-1}, {0x1.0af2f722eecb5p-1, 0x1.4d72d3a39fd01p-1},
    {0x1.09ddba6af8360p-1, 0x1.4f87a3f5026e9p-1}, {0x1.08cabb37565e2p-1, 0x1.519a4c0ba3446p-1},
    {0x1.07b9f29b8eae2p-1, 0x1.53aad05b99b7cp-1}, {0x1.06ab59c7912fbp-1, 0x1.55b9354b40bcep-1},
    {0x1.059eea0727586p-1, 0x1.57c57f336f191p-1}, {0x1.04949cc1664c5p-1, 0x1.59cfb25fae87fp-1},
    {0x1.038c6b78247fcp-1, 0x1.5bd7d30e71c73p-1}, {0x1.02864fc7729e9p-1, 0x1.5ddde57149923p-1},
    {0x1.0182436517a37p-1, 0x1.5fe1edad18919p-1}, {0x1.0080402010080p-1, 0x1.61e3efda46467p-1},
};

// Where the Gaussian source reads its two 128-entry tables from: the __constant__ originals
// (vector loads through L1), or a copy in LDS with entry i at p[i * STRIDE + OFF] (log) and
// p[(128 + i) * STRIDE + OFF] (sincos).  Kernels that also store in their main loop want LDS:
// vmcnt is shared by loads and stores, so a global table load waits for older stores.
// log(i) = {1 / c_i, -2 ln c_i}: the radius wants -2 ln u, and a factor of -2 carried by the constants (exact: a power of
// two) is one multiply per Gaussian pair less than applying it to the result — the same bits.
struct wf_tabs_global {
    __device__ __forceinline__ double2 log(int i) const { const double2 v = kWfLogTab[i]; return make_double2(v.x, -2.0 * v.y); }
    __device__ __forceinline__ double2 cis(int i) const { return kWfCisTab[i]; }
};
template <int STRIDE, int OFF>
struct wf_tabs_lds {
    const double2 *p;
    __device__ __forceinline__ double2 log(int i) const { return p[i * STRIDE + OFF]; }
    __device__ __forceinline__ double2 cis(int i) const { return p[(128 + i) * STRIDE + OFF]; }
};
template <int STRIDE, int OFF>
__device__ __forceinline__ void wf_stage_tables(double2 *p, int t, int nthreads)   // caller synchronises
{
    for (int k = t; k < 256; k += nthreads)
        p[k * STRIDE + OFF] = k < 128 ? make_double2(kWfLogTab[k].x, -2.0 * kWfLogTab[k].y) : kWfCisTab[k - 128];
}

// -2 ln((xa + 1) 2^-32): the series of the comment above with every constant scaled by -2 (exact), degree 5 (the r^6 term,
// < 1.2e-15, left out).  The result is the true value (>= 0) to ~2e-16 absolute; for u = 1 it may come out as -4e-19:
// wf_sqrt_pos floors its argument.
__device__ __forceinline__ double wf_neg2log_series(double m, int e, double2 tc)
{
    const double r = fma(m, tc.x, -1.0);
    double p = fma(r, -0.4, 0.5);
    p = fma(r, p, -2.0 / 3.0);
    p = fma(r, p, 1.0);
    p = fma(r, p, -2.0);
    const double de = (double)e;
    return fma(de, -2.0 * 6.93147180369123816490e-01, fma(r, p, tc.y) + de * (-2.0 * 1.90821492927058770002e-10));
}

// (double)xa + 1.0 for a 32-bit word, exactly, in ONE operation: 2^52 + xa is the double with xa as its low word, and
// (2^52 + xa) - (2^52 - 1) is exact (v_cvt_f64_u32 + v_add_f64 otherwise: wf_u32_plus_one_cvt, the same value — for kernels
// whose register budget has no room for the constant high word).
__device__ __forceinline__ double wf_u32_plus_one_cvt(uint32_t xa) { return (double)xa + 1.0; }
__device__ __forceinline__ double wf_u32_plus_one(uint32_t xa)
{
    return __longlong_as_double((long long)(0x4330000000000000ull | xa)) - 4503599627370495.0;
}

template <class Tabs>
__device__ __forceinline__ double wf_neg2log_unit32(uint32_t xa, const Tabs &tb)
{
    const long long ix = __double_as_longlong(wf_u32_plus_one(xa));       // 1 .. 2^32, exact
    const int e = (int)(ix >> 52) - (1023 + 32);
    const long long mant = ix & 0x000FFFFFFFFFFFFFll;
    return wf_neg2log_series(__longlong_as_double(mant | 0x3FF0000000000000ll), e, tb.log((int)(mant >> 45)));
}

// sqrt(a) for a normal-range double: v_rsq_f64 seed (relative error <= 2^-23) and one Goldschmidt step (~2^-45 = 3e-14:
// the Newton correction on the exact residual that used to follow bought the last 30 bits of a NOISE radius at two more
// operations per sample).  The argument is floored at 2^-1000 (-2 ln(u) for u = 1 once rounded to -4e-19 and
// a * rsq(tiny) turned it into 1e133): a <= 0 gives 2^-500, zero to every purpose.
__device__ __forceinline__ double wf_sqrt_pos(double a)
{
    const double af = fmax(a, 0x1.0p-1000);
    const double y = __builtin_amdgcn_rsq(af);
    const double g = af * y, h = 0.5 * y;
    return fma(g, fma(-h, g, 0.5), g);
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
        // a ^ b ^ c as ONE v_bitop3_b32 (gfx950; truth table 0x96): 2.65 cycles with three vector operands, 4.28 with the round
        // key as a scalar (the case here) — against 4.48 for two v_xor_b32 (tools/valu_probe.hip, event-time column of
        // profiles/r05_valu_probe.json); 16 fewer vector instructions per row, links - 0.4 .. - 1.2 % (r05_ab_philox_bitop3.log)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

template <class Tabs>
__device__ __forceinline__ void wf_sincos_u32(uint32_t xb, const Tabs &tb, double *sn, double *cs)   // t = xb * 2^-32
{
    const int ri = (int)(xb & 0x1FFFFFFu) - (1 << 24);
    wf_cis_sector(tb.cis((int)(xb >> 25)), (double)ri * (6.28318530717958647692 * 0x1.0p-32), sn, cs);
}

// Box-Muller from two 32-bit words: u1 = (xa + 1) 2^-32 in (0, 1], u2 = xb 2^-32 in [0, 1).
template <class Tabs>
__device__ __forceinline__ void wf_box_muller32(uint32_t xa, uint32_t xb, double sigma, const Tabs &tb, double *re, double *im)
{
    const double r = sigma * wf_sqrt_pos(wf_neg2log_unit32(xa, tb));
    double s, c;
    wf_sincos_u32(xb, tb, &s, &c);
    *re = r * c;
    *im = r * s;
}

// The two complex Gaussian samples with absolute indices 2*pair and 2*pair + 1: ONE
// Philox4x32-10 block (counter = pair index, stream id; key = seed), words (x0, x1) for the
// even sample, (x2, x3) for the odd one.  g = {re0, im0, re1, im1}.
// The Philox key schedule (20 words: seed + r * Weyl constants) is uniform and loop-invariant; left to the compiler it
// is hoisted out of a staging loop into 20 SGPRs and pushes as many other uniforms into spill lanes (v_readlane reloads
// inside the loop).  A seed made opaque at its use is re-derived there by a handful of scalar adds instead.
__device__ __forceinline__ uint64_t wf_opaque_seed(uint64_t seed)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    asm volatile("" : "+s"(k0), "+s"(k1));
    return ((uint64_t)k1 << 32) | k0;
}

template <class Tabs>
__device__ __forceinline__ void wf_gaussian_two(uint64_t pair, uint64_t stream_id, uint64_t seed, double sigma,
                                                const Tabs &tb, double g[4])
{
    const wf_philox_out p = wf_philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)stream_id,
                                             (uint32_t)(stream_id >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    wf_box_muller32(p.x0, p.x1, sigma, tb, &g[0], &g[1]);
    wf_box_muller32(p.x2, p.x3, sigma, tb, &g[2], &g[3]);
}

// wf_gaussian_two with the two Box-Muller transforms INTERLEAVED: both log-table entries are fetched together, then
// both sector entries, so the four dependent table round trips of a Philox block become two (same operations on the
// same operands as wf_gaussian_two: the same bits).  ALL_UP_FRONT: all four entries right behind the Philox rounds.
// SECTION_OFF (tools/section_budget.py ONLY — never set in a build that ships): bit 0 = the Philox rounds replaced by the
// counter's own words, bit 1 = the two Box-Muller transforms replaced by a bit cast of the words.  The tool compiles the
// front-end kernel with one section stubbed at a time and attributes the difference in the row loop's instruction count.
template <bool ALL_UP_FRONT, class Tabs, bool ONE_OP_X = true, int SECTION_OFF = 0>
__device__ __forceinline__ void wf_gaussian_two_il(uint64_t pair, uint64_t stream_id, uint64_t seed, double sigma,
                                                   const Tabs &tb, double g[4])
{
    wf_philox_out p;
    if constexpr (SECTION_OFF & 1) p = wf_philox_out{(uint32_t)pair, (uint32_t)(pair >> 32) ^ (uint32_t)seed, (uint32_t)pair ^ (uint32_t)stream_id, (uint32_t)(seed >> 32)};
    else p = wf_philox4x32_10((uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    if constexpr (SECTION_OFF & 2) {
        g[0] = __hiloint2double(0x3ff00000, (int)p.x0); g[1] = __hiloint2double(0x3ff00000, (int)p.x1);
        g[2] = __hiloint2double(0x3ff00000, (int)p.x2); g[3] = __hiloint2double(0x3ff00000, (int)p.x3);
        return;
    }
    const uint32_t xa[2] = {p.x0, p.x2}, xb[2] = {p.x1, p.x3};
    long long mant[2];
    int e[2];
    double2 tl[2], tc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const long long ix = __double_as_longlong(ONE_OP_X ? wf_u32_plus_one(xa[q]) : wf_u32_plus_one_cvt(xa[q]));
        e[q] = (int)(ix >> 52) - (1023 + 32);
        mant[q] = ix & 0x000FFFFFFFFFFFFFll;
        tl[q] = tb.log((int)(mant[q] >> 45));
    }
    if (ALL_UP_FRONT) {
#pragma unroll
        for (int q = 0; q < 2; ++q) tc[q] = tb.cis((int)(xb[q] >> 25));
    }
    double rad[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        rad[q] = sigma * wf_sqrt_pos(wf_neg2log_series(__longlong_as_double(mant[q] | 0x3FF0000000000000ll), e[q], tl[q]));
    }
    if (!ALL_UP_FRONT) {
#pragma unroll
        for (int q = 0; q < 2; ++q) tc[q] = tb.cis((int)(xb[q] >> 25));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ri = (int)(xb[q] & 0x1FFFFFFu) - (1 << 24);
        double sn, cs;
        wf_cis_sector(tc[q], (double)ri * (6.28318530717958647692 * 0x1.0p-32), &sn, &cs);
        g[2 * q] = rad[q] * cs;
        g[2 * q + 1] = rad[q] * sn;
    }
}

__device__ __forceinline__ long long wf_wave_sum_i64(long long v)
{
#pragma unroll
    for (int d = WF_WAVE / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, WF_WAVE);
    return v;
}
#endif
