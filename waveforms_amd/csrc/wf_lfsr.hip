// wf_lfsr.hip — K1: Galois LFSR PRBS by GF(2) leap-ahead.
// Replaces GLFSR.next_bit loops (reference waveforms/glfsr/glfsr.py:6-19,
// waveforms/glfsr/pn.py:98-107).  Integer path, bit-exact.
//
// One step of the register is the linear map T over GF(2)^64:
//   T e_0 = mask, T e_c = e_{c-1}.
// The context caches, per mask, T^(2^j), j = 0..63 and T^(B t), t = 0..255, B = bits per thread
// (128; column form).  Block b jumps to position skip + b*256*B with one wave doing matrix-vector products in
// parallel (lane c holds column c, XOR-reduce by DPP, result read back as a scalar); every
// thread then jumps a further t*B steps with ONE product against its own T^(B t) (the
// columns selected by the set bits of the block state, coalesced loads) and emits B bits,
// in a 32-bit register when the degree allows.  Bits are re-packed through LDS so that global
// stores are 16 B per lane, fully coalesced.
#include <vector>

#include "wf_common.h"

#define LFSR_THREADS 256
#ifndef LFSR_WORDS
#define LFSR_WORDS 2                                               // 64-bit words per thread (1, 2 or 4)
#endif
#define LFSR_BITS_PER_THREAD (64 * LFSR_WORDS)
#define LFSR_LOG2_BPT (LFSR_WORDS == 4 ? 8 : LFSR_WORDS == 2 ? 7 : 6)
#define LFSR_BITS_PER_BLOCK (LFSR_THREADS * LFSR_BITS_PER_THREAD)

static inline uint64_t host_matvec(const uint64_t *cols, uint64_t v)
{
    uint64_t y = 0;
    while (v) {
        int c = __builtin_ctzll(v);
        y ^= cols[c];
        v &= v - 1;
    }
    return y;
}

static wf_lfsr_tables *get_tables(wf_ctx *ctx, uint64_t mask)
{
    auto it = ctx->lfsr.find(mask);
    if (it != ctx->lfsr.end()) return it->second;
    wf_lfsr_tables *t = new wf_lfsr_tables();
    t->host[0][0] = mask;
    for (int c = 1; c < 64; ++c) t->host[0][c] = 1ull << (c - 1);
    for (int j = 1; j < 64; ++j)
        for (int c = 0; c < 64; ++c) t->host[j][c] = host_matvec(t->host[j - 1], t->host[j - 1][c]);
    // thread-jump table behind the 64 x 64 words: TJ[c][t] = column c of T^(B t), B = bits per thread
    std::vector<uint64_t> tj((size_t)64 * LFSR_THREADS);
    {
        uint64_t cur[64];
        for (int c = 0; c < 64; ++c) cur[c] = 1ull << c;                       // T^0
        for (int th = 0; th < LFSR_THREADS; ++th) {
            for (int c = 0; c < 64; ++c) tj[(size_t)c * LFSR_THREADS + th] = cur[c];
            for (int c = 0; c < 64; ++c) cur[c] = host_matvec(t->host[LFSR_LOG2_BPT], cur[c]);   // T^B * (.)
        }
    }
    if (hipMalloc(&t->dev, sizeof(t->host) + tj.size() * sizeof(uint64_t)) != hipSuccess ||
        hipMemcpy(t->dev, t->host, sizeof(t->host), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t->dev + 64 * 64, tj.data(), tj.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
        delete t;
        return nullptr;
    }
    ctx->lfsr[mask] = t;
    return t;
}

static uint64_t host_jump(const wf_lfsr_tables *t, uint64_t state, uint64_t steps)
{
    for (int j = 0; j < 64; ++j)
        if ((steps >> j) & 1) state = host_matvec(t->host[j], state);
    return state;
}

// XOR of a 64-bit value over the wave, returned as a wave-uniform scalar: row_shr 1, 2, 4, 8, the
// row totals pushed down by row_bcast:15 / :31, lane 63 read back (no LDS round trips, unlike the
// __shfl_xor butterfly — this product sits in a serial chain of up to 64).
__device__ __forceinline__ uint64_t lfsr_wave_xor(uint64_t v)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#define LFSR_DPP_STEP(CTRL, RM)                                                      \
    lo ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, RM, 0xf, false);   \
    hi ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, RM, 0xf, false)
    LFSR_DPP_STEP(0x111, 0xf);
    LFSR_DPP_STEP(0x112, 0xf);
    LFSR_DPP_STEP(0x114, 0xf);
    LFSR_DPP_STEP(0x118, 0xf);
    LFSR_DPP_STEP(0x142, 0xa);
    LFSR_DPP_STEP(0x143, 0xc);
#undef LFSR_DPP_STEP
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
}

// The link's SOQPSK 4-state / 2-column precoder (model.py:205-258) in the SAME launch.  That trellis keeps ONE bit per rail:
// column 0 (even symbol index) replaces state bit 1 by beta = x ^ flip, column 1 bit 0, flip = the bit being replaced
// (differential form) or 0 — so the state in front of symbol n is the running XOR of each rail's inputs, or simply the
// last two inputs.  For a whole 32768-bit block that is a LINEAR functional of the block's LFSR base state, so the host
// walks the blocks' base states (one matrix-vector product each) and hands the kernel the state in front of every block
// in its arguments: no scan, no second launch.  Inside the block a ballot gives every 16-symbol chunk its start state
// and each thread walks chunks through the trellis tables exactly as enc_emit_kernel does.
struct lfsr_emit_args {
    uint64_t start[32];          // precoder state (bit 1 | bit 0) in front of block j: bits 2 j, 2 j + 1 (1024 blocks = 3.3e7 symbols)
    uint64_t out_lo, out_hi;     // the trellis tables as enc_params carries them: output byte of table entry idx ...
    uint32_t next2;              // ... and its next state, idx = (column * 4 + state) * 2 + input
    int differential;
    int8_t *symbols;             // nullptr: PRBS only
    int map_kind;                // 0: the SOQPSK precoder above; 1 / 2: the memoryless mappers of the CPM link instead (wf_symbol_map
                                 //    kinds: 1 = bit pairs -> {-3, -1, 1, 3}, first bit the heavier; 2 = bits -> -1 / +1), no carries
};
#define LFSR_EMIT_MAX_BLOCKS 1024

__global__ __launch_bounds__(LFSR_THREADS) void lfsr_kernel(const uint64_t *__restrict__ jump,
                                                              uint64_t mask, uint64_t state,
                                                              uint64_t skip, uint8_t *__restrict__ bits,
                                                              int64_t n, int degree,
                                                              const uint64_t *__restrict__ dyn_skip,
                                                              const lfsr_emit_args E)
{
    if (dyn_skip) skip += *dyn_skip;   // stream replayed as a graph: the position lives on the device
    __shared__ uint64_t s_tab[64][64];   // all 64 jump matrices (32 KB), staged with parallel loads
    __shared__ uint64_t s_words[LFSR_THREADS * LFSR_WORDS];
    __shared__ uint64_t s_base;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const uint64_t pos0 = skip + (uint64_t)blockIdx.x * LFSR_BITS_PER_BLOCK;

    // the block jump below is a serial chain of matrix-vector products: fetching each
    // matrix from global memory inside the chain cost ~1.5 us of latency per set bit of
    // pos0, so the whole table is staged first (16 independent 16 B loads per thread)
    {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(jump);
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(&s_tab[0][0]);
#pragma unroll
        for (int k = 0; k < (64 * 64 / 2) / LFSR_THREADS; ++k) dst[k * LFSR_THREADS + t] = src[k * LFSR_THREADS + t];
    }
    __syncthreads();
    if (t < 64) {  // wave 0: block base state = T^pos0 * state
        uint64_t s = state;
        for (int j = 0; j < 64; ++j) {
            if ((pos0 >> j) & 1) {  // wave-uniform
                const uint64_t col = s_tab[j][lane];
                s = lfsr_wave_xor(((s >> lane) & 1) ? col : 0ull);
            }
        }
        if (t == 0) s_base = s;
    }
    __syncthreads();
    // thread t: T^(B t) * base — XOR of the columns picked by the set bits of the (block-uniform)
    // base state; the loads are coalesced over t
    const uint64_t base = s_base;
    const uint64_t *tj = jump + 64 * 64;
    uint64_t s = 0;
    for (int c = 0; c < degree; ++c)
        if ((base >> c) & 1) s ^= tj[c * LFSR_THREADS + t];
    if (degree <= 32) {
        uint32_t s32 = (uint32_t)s;
        const uint32_t m32 = (uint32_t)mask;
#pragma unroll 1
        for (int wi = 0; wi < 2 * LFSR_WORDS; ++wi) {
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t bit = s32 & 1u;
                s32 = (s32 >> 1) ^ (m32 & (0u - bit));
                w |= bit << k;
            }
            reinterpret_cast<uint32_t *>(s_words)[t * 2 * LFSR_WORDS + wi] = w;
        }
    } else {
#pragma unroll 1
        for (int wi = 0; wi < LFSR_WORDS; ++wi) {
            uint64_t w = 0;
#pragma unroll 16
            for (int k = 0; k < 64; ++k) {
                const uint64_t bit = s & 1;
                s = (s >> 1) ^ (mask & (0 - bit));
                w |= bit << k;
            }
            s_words[t * LFSR_WORDS + wi] = w;
        }
    }
    __syncthreads();
    if (E.symbols && E.map_kind) {
        // the CPM link's mappers (wf_encode.hip: symbol_map_kernel, parity 0) on the block's bits while they are in LDS:
        // thread t takes 16 bits per pass, chunk-interleaved so that a wave's stores are contiguous
        const int64_t blk0 = (int64_t)blockIdx.x * LFSR_BITS_PER_BLOCK;
#pragma unroll 2
        for (int r = 0; r < LFSR_BITS_PER_BLOCK / (16 * LFSR_THREADS); ++r) {
            const int p = 16 * (r * LFSR_THREADS + t);
            const int64_t gi = blk0 + p;
            if (gi >= n) break;
            const uint32_t x = (uint32_t)(s_words[p >> 6] >> (p & 63)) & 0xFFFFu;
            if (E.map_kind == 2) {
                const uint64_t lo = (((uint64_t)(x & 0xFF) * 0x0101010101010101ull) & 0x8040201008040201ull);
                const uint64_t hi = (((uint64_t)(x >> 8) * 0x0101010101010101ull) & 0x8040201008040201ull);
                const uint64_t blo = ((lo + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;   // one byte per bit: 0 / 1
                const uint64_t bhi = ((hi + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
                const uint64_t vlo = blo | ~(blo * 0xFFull), vhi = bhi | ~(bhi * 0xFFull);          // 2 b - 1: 0x01 / 0xFF
                if (gi + 16 <= n) {
                    *reinterpret_cast<ulonglong2 *>(E.symbols + gi) = make_ulonglong2(vlo, vhi);
                } else {
                    for (int k = 0; k < 16 && gi + k < n; ++k) E.symbols[gi + k] = (int8_t)(((k < 8 ? vlo : vhi) >> (8 * (k & 7))) & 0xFF);
                }
            } else {
                uint64_t v = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int val = 2 * (2 * (int)((x >> (2 * q)) & 1u) + (int)((x >> (2 * q + 1)) & 1u)) - 3;
                    v |= (uint64_t)(uint8_t)(int8_t)val << (8 * q);
                }
                const int64_t so = gi >> 1, nout = n >> 1;
                if (so + 8 <= nout) {
                    *reinterpret_cast<uint64_t *>(E.symbols + so) = v;
                } else {
                    for (int k = 0; k < 8 && so + k < nout; ++k) E.symbols[so + k] = (int8_t)((v >> (8 * k)) & 0xFF);
                }
            }
        }
    } else if (E.symbols) {
        static_assert(LFSR_WORDS == 2 && LFSR_THREADS == 256, "the precoder phase assumes 128 bits per thread, 2048 chunks per block");
        __shared__ uint8_t s_st[LFSR_BITS_PER_BLOCK / 16];      // start state of every 16-symbol chunk
        __shared__ unsigned s_wp[LFSR_THREADS / 64];
        const unsigned carry = (unsigned)((E.start[blockIdx.x >> 5] >> (2 * (blockIdx.x & 31))) & 3ull);
        auto bit_at = [&](int p) { return (unsigned)((s_words[p >> 6] >> (p & 63)) & 1ull); };
        if (E.differential) {
            const uint64_t w0 = s_words[t * 2], w1 = s_words[t * 2 + 1];
            const bool pe = ((__popcll(w0 & 0x5555555555555555ull) + __popcll(w1 & 0x5555555555555555ull)) & 1) != 0;
            const bool po = ((__popcll(w0 & 0xAAAAAAAAAAAAAAAAull) + __popcll(w1 & 0xAAAAAAAAAAAAAAAAull)) & 1) != 0;
            const unsigned long long be = __builtin_amdgcn_ballot_w64(pe), bo = __builtin_amdgcn_ballot_w64(po);
            const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            unsigned st = ((unsigned)(__popcll(be & lt) & 1) << 1) | (unsigned)(__popcll(bo & lt) & 1);   // rails before this thread, inside the wave
            if (lane == 0) s_wp[t >> 6] = ((unsigned)(__popcll(be) & 1) << 1) | (unsigned)(__popcll(bo) & 1);
            __syncthreads();
            for (int w = 0; w < (t >> 6); ++w) st ^= s_wp[w];
            st ^= carry;                                         // (bit 1: even positions = column 0, bit 0: odd positions)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s_st[8 * t + c] = (uint8_t)st;
                const unsigned ch = (unsigned)(((c < 4 ? w0 : w1) >> (16 * (c & 3))) & 0xFFFFull);
                st ^= ((unsigned)(__popc(ch & 0x5555u) & 1) << 1) | (unsigned)(__popc(ch & 0xAAAAu) & 1);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int p = 128 * t + 16 * c;                  // the two inputs before the chunk: position p - 2 is even (column 0: bit 1)
                s_st[8 * t + c] = (uint8_t)(p == 0 ? carry : ((bit_at(p - 2) << 1) | bit_at(p - 1)));
            }
        }
        __syncthreads();
        const int64_t blk0 = (int64_t)blockIdx.x * LFSR_BITS_PER_BLOCK;
#pragma unroll 2
        for (int r = 0; r < LFSR_BITS_PER_BLOCK / (16 * LFSR_THREADS); ++r) {
            const int c = r * LFSR_THREADS + t, p = 16 * c;
            const int64_t gi = blk0 + p;
            if (gi >= n) break;
            const unsigned x = (unsigned)((s_words[p >> 6] >> (p & 63)) & 0xFFFFull);
            unsigned st = s_st[c];
            uint64_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned idx = (((unsigned)(k & 1) * 4u + st) << 1) | ((x >> k) & 1u);
                const uint64_t o = ((idx < 8 ? E.out_lo : E.out_hi) >> (8 * (idx & 7))) & 0xFFull;
                if (k < 8) lo |= o << (8 * k); else hi |= o << (8 * (k - 8));
                st = (E.next2 >> (2 * idx)) & 3u;
            }
            if (gi + 16 <= n) {
                *reinterpret_cast<ulonglong2 *>(E.symbols + gi) = make_ulonglong2(lo, hi);
            } else {
                for (int k = 0; k < 16 && gi + k < n; ++k) E.symbols[gi + k] = (int8_t)(((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xFF);
            }
        }
    }

    const int64_t blk_byte0 = (int64_t)blockIdx.x * LFSR_BITS_PER_BLOCK;
#pragma unroll 4
    for (int r = 0; r < LFSR_BITS_PER_BLOCK / (LFSR_THREADS * 16); ++r) {
        const int g = r * (LFSR_THREADS * 16) + t * 16;  // byte (= bit) offset inside the block
        const int64_t gi = blk_byte0 + g;
        if (gi >= n) break;
        const uint32_t x = (uint32_t)(s_words[g >> 6] >> (g & 63)) & 0xFFFFu;
        // spread 8 bits to 8 bytes (LSB first): classic multiply-mask trick
        const uint64_t lo = (((uint64_t)(x & 0xFF) * 0x0101010101010101ull) & 0x8040201008040201ull);
        const uint64_t hi = (((uint64_t)(x >> 8) * 0x0101010101010101ull) & 0x8040201008040201ull);
        const uint64_t blo = ((lo + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
        const uint64_t bhi = ((hi + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
        if (gi + 16 <= n) {
            *reinterpret_cast<ulonglong2 *>(bits + gi) = make_ulonglong2(blo, bhi);
        } else {
            for (int k = 0; k < 16 && gi + k < n; ++k)
                bits[gi + k] = (uint8_t)(((k < 8 ? blo : bhi) >> (8 * (k & 7))) & 0xFF);
        }
    }
}


extern "C" int wf_lfsr_generate(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state,
                                uint64_t skip, uint8_t *d_bits, int64_t n, uint64_t *h_state_out,
                                void *stream)
{
    return wf_lfsr_generate_dyn(ctx, degree, mask, state, skip, nullptr, d_bits, n, h_state_out, stream);
}

int wf_lfsr_generate_dyn(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip,
                         const uint64_t *d_dyn_skip, uint8_t *d_bits, int64_t n, uint64_t *h_state_out, void *stream,
                         const lfsr_emit_args *emit)
{
    WF_REQUIRE(ctx != nullptr, "wf_lfsr_generate: ctx is NULL");
    if (degree < 2 || degree > 64) {
        wf_set_error("PRBS Polynomial Not Defined for %d.", degree);
        return WF_ERR_KEY;
    }
    WF_REQUIRE(n >= 0, "wf_lfsr_generate: n = %lld", (long long)n);
    WF_REQUIRE(degree == 64 || ((mask >> degree) == 0 && (state >> degree) == 0),
               "wf_lfsr_generate: mask/state wider than the register (degree %d)", degree);
    WF_REQUIRE(n == 0 || (d_bits && (reinterpret_cast<uintptr_t>(d_bits) & 15) == 0),
               "wf_lfsr_generate: d_bits must be non-NULL and 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    wf_lfsr_tables *t = get_tables(ctx, mask);
    if (!t) {
        wf_set_error("wf_lfsr_generate: could not build jump tables");
        return WF_ERR_NOMEM;
    }
    if (h_state_out) *h_state_out = host_jump(t, state, skip + (uint64_t)n);
    if (n == 0) return WF_OK;
    const int64_t blocks = (n + LFSR_BITS_PER_BLOCK - 1) / LFSR_BITS_PER_BLOCK;
    WF_REQUIRE(blocks < (1ll << 31), "wf_lfsr_generate: n too large for one launch");
    hipLaunchKernelGGL(lfsr_kernel, dim3((unsigned)blocks), dim3(LFSR_THREADS), 0, wf_stream(stream),
                       t->dev, mask, state, skip, d_bits, n, degree, d_dyn_skip, emit ? *emit : lfsr_emit_args{});
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// PRBS bits AND the CPM link's memoryless symbol mapper (wf_symbol_map kinds 1 / 2 at parity 0) in one launch.
// Returns 1 — not an error — for any other mapper: the caller runs wf_lfsr_generate + wf_symbol_map.
int wf_lfsr_generate_map(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip, uint8_t *d_bits, int64_t nbits,
                         int map_kind, int8_t *d_symbols, void *stream)
{
    if ((map_kind != 1 && map_kind != 2) || nbits < 1 || (map_kind == 1 && (nbits & 1))) return 1;
    WF_REQUIRE(d_symbols && (reinterpret_cast<uintptr_t>(d_symbols) & 15) == 0, "wf_lfsr_generate_map: d_symbols must be non-NULL and 16-byte aligned");
    lfsr_emit_args E{};
    E.symbols = d_symbols;
    E.map_kind = map_kind;
    return wf_lfsr_generate_dyn(ctx, degree, mask, state, skip, nullptr, d_bits, nbits, nullptr, stream, &E);
}

// Functionals of one 32768-bit block as seen from its base state (linear over GF(2)): parity of the output bits at
// even / odd positions, the last bit but one, the last bit.  Bit c of a functional = its value for base state e_c.
static void lfsr_block_functionals(wf_lfsr_tables *t, uint64_t mask)
{
    if (t->f_ready) return;
    for (int c = 0; c < 64; ++c) {
        uint64_t s = 1ull << c;
        unsigned pe = 0, po = 0, m2 = 0, m1 = 0;
        for (int k = 0; k < LFSR_BITS_PER_BLOCK; ++k) {
            const unsigned bit = (unsigned)(s & 1ull);
            s = (s >> 1) ^ (mask & (0ull - (uint64_t)bit));
            if (k & 1) po ^= bit; else pe ^= bit;
            if (k == LFSR_BITS_PER_BLOCK - 2) m2 = bit;
            if (k == LFSR_BITS_PER_BLOCK - 1) m1 = bit;
        }
        t->f_even |= (uint64_t)pe << c;
        t->f_odd |= (uint64_t)po << c;
        t->f_m2 |= (uint64_t)m2 << c;
        t->f_m1 |= (uint64_t)m1 << c;
    }
    t->f_ready = true;
}

// PRBS bits AND the link's SOQPSK precoder symbols in one launch (see lfsr_emit_args).  h_next / h_out: the trellis
// tables [2 columns][4 states][2 inputs]; they must BE that trellis, in its differential or its plain form.
// Returns 1 — not an error — for any other trellis or a burst of more than LFSR_EMIT_MAX_BLOCKS blocks (3.3e7
// symbols): the caller then runs wf_lfsr_generate + wf_fsm_encode.
int wf_soqpsk_prbs_encode(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip, const uint8_t *h_next,
                          const int8_t *h_out, uint8_t *d_bits, int64_t n, int8_t *d_symbols, void *stream, void *mid_event)
{
    WF_REQUIRE(ctx && h_next && h_out, "wf_soqpsk_prbs_encode: NULL argument");
    const int64_t blocks = (n + LFSR_BITS_PER_BLOCK - 1) / LFSR_BITS_PER_BLOCK;
    if (n < 1 || blocks > LFSR_EMIT_MAX_BLOCKS || degree < 2 || degree > 64) return 1;
    int diff = -1;
    for (int d = 0; d < 2 && diff < 0; ++d) {
        bool ok = true;
        for (int c = 0; c < 2 && ok; ++c)
            for (int s = 0; s < 4 && ok; ++s)
                for (int x = 0; x < 2 && ok; ++x) {
                    const int flip = d ? (c == 0 ? (s >> 1) : (s & 1)) : 0, beta = x ^ flip;
                    ok = h_next[(c * 4 + s) * 2 + x] == (c == 0 ? (s & 1) + 2 * beta : (s & 2) + beta);
                }
        if (ok) diff = d;
    }
    if (diff < 0) return 1;
    WF_REQUIRE(d_symbols && (reinterpret_cast<uintptr_t>(d_symbols) & 15) == 0, "wf_soqpsk_prbs_encode: d_symbols must be non-NULL and 16-byte aligned");
    WF_REQUIRE(degree == 64 || ((mask >> degree) == 0 && (state >> degree) == 0), "wf_soqpsk_prbs_encode: mask/state wider than the register");
    WF_HIP(hipSetDevice(ctx->device));
    wf_lfsr_tables *t = get_tables(ctx, mask);
    if (!t) {
        wf_set_error("wf_soqpsk_prbs_encode: could not build jump tables");
        return WF_ERR_NOMEM;
    }
    lfsr_block_functionals(t, mask);
    lfsr_emit_args E{};
    for (int k = 0; k < 16; ++k) {
        E.next2 |= (uint32_t)(h_next[k] & 3) << (2 * k);
        (k < 8 ? E.out_lo : E.out_hi) |= (uint64_t)(uint8_t)h_out[k] << (8 * (k & 7));
    }
    E.differential = diff;
    E.symbols = d_symbols;
    // the precoder state in front of every block, from the blocks' base states (base_{j+1} = T^32768 base_j)
    uint64_t base = host_jump(t, state, skip);
    unsigned st = 0, rails = 0;                                      // link: encoder starts in state 0 at an even symbol index
    static_assert(LFSR_BITS_PER_BLOCK == 32768, "T^(block) is table 15");
    for (int64_t j = 0; j < blocks; ++j) {
        E.start[j >> 5] |= (uint64_t)st << (2 * (j & 31));
        if (diff) {
            rails ^= ((unsigned)(__builtin_popcountll(t->f_even & base) & 1) << 1) | (unsigned)(__builtin_popcountll(t->f_odd & base) & 1);
            st = rails;
        } else {
            st = ((unsigned)(__builtin_popcountll(t->f_m2 & base) & 1) << 1) | (unsigned)(__builtin_popcountll(t->f_m1 & base) & 1);
        }
        base = host_matvec(t->host[15], base);
    }
    const int rc = wf_lfsr_generate_dyn(ctx, degree, mask, state, skip, nullptr, d_bits, n, nullptr, stream, &E);
    if (rc) return rc;
    if (mid_event) WF_HIP(hipEventRecord(static_cast<hipEvent_t>(mid_event), wf_stream(stream)));   // (the link's stage timing: PRBS + precoder | nothing)
    return WF_OK;
}
