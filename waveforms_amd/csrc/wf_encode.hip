// wf_encode.hip — K2: bits -> symbols.
//  * wf_fsm_encode: TrellisEncoder.encode (reference waveforms/cpm/trellis/encoder.py:17-48)
//    for any trellis (<= 16 states, <= 4 bits/symbol, any number of columns) as a
//    prefix scan over state-transition maps: a run of symbols is a function
//    state -> state, packed 16 x 4 bit in one u64; composition is associative, so
//    thread maps are scanned per wave (shuffles), per block (LDS) and across blocks
//    (one small kernel), then every thread re-walks its 16 symbols from its now known
//    start state and emits them.  Integer path, bit-exact.
//  * wf_symbol_map: the element-wise precoder / mappers (a2').
#include <string.h>

#include "wf_common.h"

#define ENC_THREADS 256
#define ENC_SYM_PER_THREAD 16
#define ENC_SYM_PER_BLOCK (ENC_THREADS * ENC_SYM_PER_THREAD)

struct enc_params {
    int columns, states, card, ninp;
    int col0;  // i0 % columns
    int state0;
    int64_t nsym;
    // SMALL trellises (<= 4 states, <= 16 table entries — the SOQPSK 4x2 pair): both tables ride in
    // registers, 2 bits per next-state entry and one byte per output symbol.  From LDS the 16
    // entries sit in 4 banks and 64 lanes of random lookups serialise on them.
    uint32_t next2;
    uint64_t out_lo, out_hi;
};

__device__ __forceinline__ uint64_t map_identity()
{
    return 0xFEDCBA9876543210ull;
}

// h = g after f  (apply f first).  NS = number of states the maps are defined on (nibbles
// above NS stay identity-free zeros and are never read).
template <int NS>
__device__ __forceinline__ uint64_t map_compose(uint64_t f, uint64_t g)
{
    uint64_t h = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const unsigned fs = (unsigned)(f >> (4 * s)) & 15u;
        h |= ((g >> (4 * fs)) & 15ull) << (4 * s);
    }
    return h;
}

// Value of the lane `CTRL` says (DPP), identity map where there is none.  Only the words that
// hold states < NS travel.
template <int NS, int CTRL, int RM>
__device__ __forceinline__ uint64_t map_dpp(uint64_t v)
{
    const uint64_t id = map_identity();
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)id, (int)(uint32_t)v, CTRL, RM, 0xf, false);
    uint32_t hi = (uint32_t)(id >> 32);
    if (NS > 8) hi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)(uint32_t)(v >> 32), CTRL, RM, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}

// Inclusive wave scan of maps in lane order (compose earlier lanes first): Hillis-Steele inside
// rows of 16 lanes, then the row totals are pushed into the following rows.  VALU only.
template <int NS>
__device__ __forceinline__ uint64_t map_wave_scan(uint64_t inc)
{
    inc = map_compose<NS>(map_dpp<NS, 0x111, 0xf>(inc), inc);
    inc = map_compose<NS>(map_dpp<NS, 0x112, 0xf>(inc), inc);
    inc = map_compose<NS>(map_dpp<NS, 0x114, 0xf>(inc), inc);
    inc = map_compose<NS>(map_dpp<NS, 0x118, 0xf>(inc), inc);
    inc = map_compose<NS>(map_dpp<NS, 0x142, 0xa>(inc), inc);
    inc = map_compose<NS>(map_dpp<NS, 0x143, 0xc>(inc), inc);
    return inc;
}

// 8 bytes of 0 / 1 -> 8 nibbles (byte k -> nibble k)
__device__ __forceinline__ uint32_t bytes_to_nibbles(uint64_t x)
{
    uint64_t t = x & 0x0101010101010101ull;
    t = (t | (t >> 4)) & 0x0011001100110011ull;
    t = (t | (t >> 8)) & 0x0000111100001111ull;
    t = (t | (t >> 16)) & 0x0000000011111111ull;
    return (uint32_t)t;
}

// Gather the input values of this thread's 16 symbols (4 bits each) into one u64.
__device__ __forceinline__ uint64_t load_inputs(const uint8_t *__restrict__ bits, int64_t sym0,
                                                int64_t nsym, int card, int *nvalid)
{
    int64_t rem = nsym - sym0;
    const int nv = rem >= ENC_SYM_PER_THREAD ? ENC_SYM_PER_THREAD : (rem > 0 ? (int)rem : 0);
    *nvalid = nv;
    uint64_t packed = 0;
    const uint8_t *p = bits + sym0 * card;
    if (nv == ENC_SYM_PER_THREAD && card == 1) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
        return (uint64_t)bytes_to_nibbles(v.x) | ((uint64_t)bytes_to_nibbles(v.y) << 32);
    }
    if (nv == ENC_SYM_PER_THREAD) {
        // card * 16 bytes, 16-byte aligned (sym0 is a multiple of 16)
        for (int q = 0; q < card; ++q) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p + 16 * q);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned bit = (unsigned)(((k < 8 ? v.x : v.y) >> (8 * (k & 7))) & 1ull);
                const int bi = q * 16 + k;  // bit index inside the thread's run
                const int sym = bi / card, x = bi % card;
                packed |= (uint64_t)bit << (4 * sym + (card - x - 1));
            }
        }
    } else {
        for (int sym = 0; sym < nv; ++sym)
            for (int x = 0; x < card; ++x)
                packed |= (uint64_t)(p[sym * card + x] & 1) << (4 * sym + (card - x - 1));
    }
    return packed;
}

// Kernel A: per-thread maps, block-exclusive prefix maps, block aggregates.
template <int NS, bool SMALL>
__global__ __launch_bounds__(ENC_THREADS) void enc_reduce_kernel(
    const uint8_t *__restrict__ bits, const uint8_t *__restrict__ tab_next, enc_params P,
    uint64_t *__restrict__ thread_excl, uint64_t *__restrict__ block_agg)
{
    __shared__ uint8_t s_next[1024];
    __shared__ uint64_t s_wave[ENC_THREADS / WF_WAVE];
    const int t = threadIdx.x;
    __shared__ uint8_t s_T[SMALL ? 1024 : 4];
    if (!SMALL) {
        const int tabn = P.columns * P.states * P.ninp;
        for (int k = t; k < tabn; k += ENC_THREADS) s_next[k] = tab_next[k];
    } else {
        // SMALL: a run's map fits 8 bits (4 states x 2 bits) and a symbol has <= 4 (column, input)
        // kinds, so "compose the run so far with one more symbol" is a 1 KB table: ONE byte lookup
        // per symbol instead of walking every start state.
        for (int e = t; e < 1024; e += ENC_THREADS) {
            const int cur = e >> 2, sel = e & 3;
            const int col = sel / P.ninp, inp = sel - col * P.ninp;
            int r = 0;
            for (int s0 = 0; s0 < 4; ++s0) {
                const int st = (cur >> (2 * s0)) & 3;
                const int idx = (col * P.states + st) * P.ninp + inp;
                const int nx = (st < P.states && col < P.columns) ? (int)((P.next2 >> (2 * idx)) & 3u) : st;
                r |= nx << (2 * s0);
            }
            s_T[e] = (uint8_t)r;
        }
    }
    __syncthreads();

    const int64_t gthread = (int64_t)blockIdx.x * ENC_THREADS + t;
    const int64_t sym0 = gthread * ENC_SYM_PER_THREAD;
    int nv;
    const uint64_t inps = load_inputs(bits, sym0, P.nsym, P.card, &nv);
    const int colstart = (int)((P.col0 + sym0) % P.columns);

    uint64_t m = 0;
    if (SMALL) {
        int cur = 0xE4, col = colstart;   // identity: state s -> s
        const uint32_t ilo = (uint32_t)inps, ihi = (uint32_t)(inps >> 32);
        for (int k = 0; k < nv; ++k) {
            const int inp = (int)(((k < 8 ? ilo : ihi) >> (4 * (k & 7))) & 15u);
            cur = s_T[cur * 4 + col * P.ninp + inp];
            col = col + 1 == P.columns ? 0 : col + 1;
        }
#pragma unroll
        for (int s0 = 0; s0 < 4; ++s0) m |= (uint64_t)((cur >> (2 * s0)) & 3) << (4 * s0);
    } else
#pragma unroll
    for (int s0 = 0; s0 < NS; ++s0) {
        int st = s0;
        if (s0 < P.states) {
            int col = colstart;
            for (int k = 0; k < nv; ++k) {
                const int inp = (int)((inps >> (4 * k)) & 15ull);
                const int idx = (col * P.states + st) * P.ninp + inp;
                st = SMALL ? (int)((P.next2 >> (2 * idx)) & 3u) : (int)s_next[idx];
                col = col + 1 == P.columns ? 0 : col + 1;
            }
        }
        m |= (uint64_t)st << (4 * s0);
    }
    // wave inclusive scan (compose in thread order)
    const int lane = t & 63, wave = t >> 6;
    const uint64_t inc = map_wave_scan<NS>(m);
    if (lane == 63) s_wave[wave] = inc;
    const uint64_t excl = map_dpp<NS, 0x138, 0xf>(inc);   // wave_shr:1, identity into lane 0
    __syncthreads();
    uint64_t pre = map_identity();
    for (int w = 0; w < wave; ++w) pre = map_compose<NS>(pre, s_wave[w]);
    thread_excl[gthread] = map_compose<NS>(pre, excl);
    if (t == ENC_THREADS - 1) block_agg[blockIdx.x] = map_compose<NS>(pre, inc);
}

// Kernel B: start state of every block (one workgroup; threads own contiguous runs).
#define ENC_SCAN_THREADS 1024
template <int NS>
__global__ __launch_bounds__(ENC_SCAN_THREADS) void enc_block_scan_kernel(const uint64_t *__restrict__ block_agg,
                                                                           int nblocks, int state0,
                                                                           const int *__restrict__ state_in,
                                                                           uint8_t *__restrict__ block_state)
{
    if (state_in) state0 = *state_in & 15;   // streaming: encoder state carried on the device
    __shared__ uint64_t s_wave[ENC_SCAN_THREADS / WF_WAVE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (nblocks + ENC_SCAN_THREADS - 1) / ENC_SCAN_THREADS;
    const int b0 = min(nblocks, t * per), b1 = min(nblocks, b0 + per);
    uint64_t run = map_identity();
    for (int b = b0; b < b1; ++b) run = map_compose<NS>(run, block_agg[b]);
    const uint64_t inc = map_wave_scan<NS>(run);
    if (lane == 63) s_wave[wave] = inc;
    uint64_t excl = map_dpp<NS, 0x138, 0xf>(inc);
    __syncthreads();
    uint64_t pre = map_identity();
    for (int w = 0; w < wave; ++w) pre = map_compose<NS>(pre, s_wave[w]);
    excl = map_compose<NS>(pre, excl);
    int st = (int)((excl >> (4 * state0)) & 15ull);
    for (int b = b0; b < b1; ++b) {
        block_state[b] = (uint8_t)st;
        st = (int)((block_agg[b] >> (4 * st)) & 15ull);
    }
}

// Kernel C: emit symbols.
template <bool SMALL>
__global__ __launch_bounds__(ENC_THREADS) void enc_emit_kernel(
    const uint8_t *__restrict__ bits, const uint8_t *__restrict__ tab_next,
    const int8_t *__restrict__ tab_out, enc_params P, const uint64_t *__restrict__ thread_excl,
    const uint8_t *__restrict__ block_state, int8_t *__restrict__ symbols, int *__restrict__ final_state,
    int *__restrict__ state_at, int64_t at_index)
{
    __shared__ uint8_t s_next[1024];
    __shared__ int8_t s_out[1024];
    const int t = threadIdx.x;
    if (!SMALL) {
        const int tabn = P.columns * P.states * P.ninp;
        for (int k = t; k < tabn; k += ENC_THREADS) {
            s_next[k] = tab_next[k];
            s_out[k] = tab_out[k];
        }
        __syncthreads();
    }
    const int64_t gthread = (int64_t)blockIdx.x * ENC_THREADS + t;
    const int64_t sym0 = gthread * ENC_SYM_PER_THREAD;
    int nv;
    const uint64_t inps = load_inputs(bits, sym0, P.nsym, P.card, &nv);
    if (nv == 0) return;
    int st = (int)((thread_excl[gthread] >> (4 * block_state[blockIdx.x])) & 15ull);
    int col = (int)((P.col0 + sym0) % P.columns);
    if (state_at && sym0 == at_index) *state_at = st;   // encoder state BEFORE symbol at_index
    uint64_t lo = 0, hi = 0;
    for (int k = 0; k < nv; ++k) {
        const int inp = (int)((inps >> (4 * k)) & 15ull);
        const int idx = (col * P.states + st) * P.ninp + inp;
        const uint64_t o = SMALL ? (((idx < 8 ? P.out_lo : P.out_hi) >> (8 * (idx & 7))) & 0xFFull) : (uint64_t)(uint8_t)s_out[idx];
        if (k < 8) lo |= o << (8 * k); else hi |= o << (8 * (k - 8));
        st = SMALL ? (int)((P.next2 >> (2 * idx)) & 3u) : (int)s_next[idx];
        col = col + 1 == P.columns ? 0 : col + 1;
    }
    if (nv == ENC_SYM_PER_THREAD) {
        *reinterpret_cast<ulonglong2 *>(symbols + sym0) = make_ulonglong2(lo, hi);
    } else {
        for (int k = 0; k < nv; ++k)
            symbols[sym0 + k] = (int8_t)(((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xFF);
    }
    if (sym0 + nv == P.nsym) {
        *final_state = st;
        if (state_at && at_index == P.nsym) *state_at = st;
    }
}

int wf_fsm_encode_core(wf_ctx *ctx, const uint8_t *h_next, const int8_t *h_out, int columns, int states, int card,
                       const uint8_t *d_bits, int64_t nbits, int64_t i0, int state0, const int *d_state_in,
                       int8_t *d_symbols, int *h_state_out, int *d_state_at, int64_t at_index, void *stream)
{
    WF_REQUIRE(ctx && h_next && h_out, "wf_fsm_encode: NULL argument");
    WF_REQUIRE(columns >= 1 && states >= 1 && states <= 16 && card >= 1 && card <= 4,
               "wf_fsm_encode: unsupported trellis (columns %d states %d card %d)", columns, states, card);
    const int ninp = 1 << card;
    const int tabn = columns * states * ninp;
    WF_REQUIRE(tabn <= 1024, "wf_fsm_encode: trellis table too large (%d)", tabn);
    WF_REQUIRE(state0 >= 0 && state0 < states && i0 >= 0, "wf_fsm_encode: bad i/state");
    // a next-state entry outside the trellis would index the state maps / LDS tables out of bounds
    for (int k = 0; k < tabn; ++k)
        WF_REQUIRE(h_next[k] < states, "wf_fsm_encode: next-state table entry %d is %d (states %d)", k, (int)h_next[k], states);
    if (nbits % card) {
        wf_set_error("Input length must be a multiple of FSM cardinality.");
        return WF_ERR_VALUE;
    }
    const int64_t nsym = nbits / card;
    if (h_state_out) *h_state_out = state0;
    if (nsym == 0) return WF_OK;
    WF_REQUIRE(d_bits && d_symbols && (reinterpret_cast<uintptr_t>(d_bits) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_symbols) & 15) == 0,
               "wf_fsm_encode: device pointers must be non-NULL and 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    hipStream_t s = wf_stream(stream);
    const int64_t nblocks = (nsym + ENC_SYM_PER_BLOCK - 1) / ENC_SYM_PER_BLOCK;
    WF_REQUIRE(nblocks < (1ll << 30), "wf_fsm_encode: input too large for one launch");
    // scratch: thread maps (nblocks*256) + block aggregates (nblocks) + block states (bytes)
    const size_t words = (size_t)nblocks * ENC_THREADS + (size_t)nblocks + ((size_t)nblocks + 7) / 8 + 8;
    int rc = wf_ctx_reserve_fsm(ctx, words);
    if (rc) return rc;
    uint64_t *thread_excl = ctx->d_fsm_scratch;
    uint64_t *block_agg = thread_excl + (size_t)nblocks * ENC_THREADS;
    uint8_t *block_state = reinterpret_cast<uint8_t *>(block_agg + nblocks);
    // tables: next at d_tables[0..1023], out at d_tables[1024..2047]
    // the tables rarely change between calls: upload only when they differ from what the
    // context already holds on the device
    if (ctx->tables_cached != tabn || memcmp(ctx->h_tables_cache, h_next, tabn) != 0 ||
        memcmp(ctx->h_tables_cache + 1024, h_out, tabn) != 0) {
        memcpy(ctx->h_tables_cache, h_next, tabn);
        memcpy(ctx->h_tables_cache + 1024, h_out, tabn);
        ctx->tables_cached = tabn;
        // The cache is per context and the upload is ordered on THIS call's stream only, so an event is
        // recorded behind it; a later call on another stream waits for that event (below) instead of the
        // host blocking here on everything queued on the stream.  (The upload itself is a copy from
        // pageable memory: the FIRST call of a context with a given table set must not be made inside a
        // hipGraph capture; later calls — cache hits — may.)
        WF_HIP(hipMemcpyAsync(ctx->d_tables, ctx->h_tables_cache, 2048, hipMemcpyHostToDevice, s));
        if (!ctx->tables_event) WF_HIP(hipEventCreateWithFlags(&ctx->tables_event, hipEventDisableTiming));
        WF_HIP(hipEventRecord(ctx->tables_event, s));
        ctx->tables_stream = s;
        ctx->tables_pending = true;
    } else if (ctx->tables_pending) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap == hipStreamCaptureStatusNone) {            // (inside a capture nothing may be queried or waited for: the caller has
            if (hipEventQuery(ctx->tables_event) == hipSuccess) ctx->tables_pending = false;   //  synchronised since the upload — see above)
            else if (s != ctx->tables_stream) WF_HIP(hipStreamWaitEvent(s, ctx->tables_event, 0));
        }
    }
    enc_params P{columns, states, card, ninp, (int)(i0 % columns), state0, nsym, 0u, 0ull, 0ull};
    const bool small = states <= 4 && tabn <= 16 && columns * ninp <= 4;
    if (small)
        for (int k = 0; k < tabn; ++k) {
            P.next2 |= (uint32_t)(h_next[k] & 3) << (2 * k);
            (k < 8 ? P.out_lo : P.out_hi) |= (uint64_t)(uint8_t)h_out[k] << (8 * (k & 7));
        }
#define ENC_A(NS, SM) hipLaunchKernelGGL((enc_reduce_kernel<NS, SM>), dim3((unsigned)nblocks), dim3(ENC_THREADS), 0, s, d_bits, \
                                        ctx->d_tables, P, thread_excl, block_agg)
    if (small) ENC_A(4, true); else if (states <= 4) ENC_A(4, false); else if (states <= 8) ENC_A(8, false); else ENC_A(16, false);
#undef ENC_A
    WF_LAUNCH_CHECK();
    WF_REQUIRE(!d_state_at || (at_index >= 0 && at_index <= nsym && at_index % ENC_SYM_PER_THREAD == 0),
               "wf_fsm_encode: carry index must be a multiple of %d inside the block", ENC_SYM_PER_THREAD);
#define ENC_B(NS) hipLaunchKernelGGL(enc_block_scan_kernel<NS>, dim3(1), dim3(ENC_SCAN_THREADS), 0, s, block_agg, \
                                    (int)nblocks, state0, d_state_in, block_state)
    if (states <= 4) ENC_B(4); else if (states <= 8) ENC_B(8); else ENC_B(16);
#undef ENC_B
    WF_LAUNCH_CHECK();
    if (small)
        hipLaunchKernelGGL(enc_emit_kernel<true>, dim3((unsigned)nblocks), dim3(ENC_THREADS), 0, s, d_bits,
                           ctx->d_tables, reinterpret_cast<const int8_t *>(ctx->d_tables + 1024), P,
                           thread_excl, block_state, d_symbols, ctx->d_small, d_state_at, at_index);
    else
        hipLaunchKernelGGL(enc_emit_kernel<false>, dim3((unsigned)nblocks), dim3(ENC_THREADS), 0, s, d_bits,
                           ctx->d_tables, reinterpret_cast<const int8_t *>(ctx->d_tables + 1024), P,
                           thread_excl, block_state, d_symbols, ctx->d_small, d_state_at, at_index);
    WF_LAUNCH_CHECK();
    if (h_state_out) {
        WF_HIP(hipMemcpyAsync(ctx->h_small, ctx->d_small, sizeof(int), hipMemcpyDeviceToHost, s));
        WF_HIP(hipStreamSynchronize(s));
        *h_state_out = ctx->h_small[0];
    }
    return WF_OK;
}

extern "C" int wf_fsm_encode(wf_ctx *ctx, const uint8_t *h_next, const int8_t *h_out, int columns,
                             int states, int card, const uint8_t *d_bits, int64_t nbits, int64_t i0,
                             int state0, int8_t *d_symbols, int *h_state_out, void *stream)
{
    return wf_fsm_encode_core(ctx, h_next, h_out, columns, states, card, d_bits, nbits, i0, state0, nullptr,
                              d_symbols, h_state_out, nullptr, 0, stream);
}

// ------------------------------------------------------------------ a2' mappers
__global__ void symbol_map_kernel(int kind, const uint8_t *__restrict__ bits, int64_t n, int parity,
                                  int mem0, int mem1, int8_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nout = kind == 1 ? n / 2 : n;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nout; k += stride) {
        int v;
        if (kind == 0) {
            // a = mem ++ bits; out[k] = sign_k * (2 a[k+1] - 1) * (a[k] - a[k+2]);
            // sign is -1 where (k - parity) is even and k >= parity (i_arr[i::2] = -1)
            const int a0 = k >= 2 ? bits[k - 2] : (k == 0 ? mem0 : mem1);
            const int a1 = k >= 1 ? bits[k - 1] : mem1;
            const int a2 = bits[k];
            const int sign = (k >= parity && ((k - parity) & 1) == 0) ? -1 : 1;
            v = sign * (2 * a1 - 1) * (a0 - a2);
        } else if (kind == 1) {
            // 2*(2*bits[i::2] + bits[(i+1)%2::2]) - 3 with i = parity (already updated)
            const int hi = bits[2 * k + parity];
            const int lo = bits[2 * k + ((parity + 1) & 1)];
            v = 2 * (2 * hi + lo) - 3;
        } else {
            v = 2 * (int)bits[k] - 1;
        }
        out[k] = (int8_t)v;
    }
}

extern "C" int wf_symbol_map(wf_ctx *ctx, int kind, const uint8_t *d_bits, int64_t n, int parity,
                             int mem0, int mem1, int8_t *d_symbols, void *stream)
{
    WF_REQUIRE(ctx && kind >= 0 && kind <= 2 && n >= 0, "wf_symbol_map: bad argument");
    if (kind == 1 && (n & 1)) {
        wf_set_error("Odd length bit array passed into quaternary mapper.");
        return WF_ERR_VALUE;
    }
    if (n == 0) return WF_OK;
    WF_REQUIRE(d_bits && d_symbols, "wf_symbol_map: NULL device pointer");
    WF_REQUIRE(parity == 0 || parity == 1, "wf_symbol_map: parity must be 0 or 1");
    WF_HIP(hipSetDevice(ctx->device));
    const int grid = wf_grid_for(n, 256, 4096);
    hipLaunchKernelGGL(symbol_map_kernel, dim3(grid), dim3(256), 0, wf_stream(stream), kind, d_bits,
                       n, parity, mem0, mem1, d_symbols);
    WF_LAUNCH_CHECK();
    return WF_OK;
}
