// wf_cpm_wide.hip — the generic CPM trellis detector for trellises of 17 .. 64 states: lane = state, one WAVE = one
// detector (wf_cpm_detect.hip: one 16-lane DPP row = one detector, 4 per wave; wf_cpm_lanes.hip: one lane = one detector).
//
// What it is for: the 64-state design of ARTM multi-h CPM (Lp = 2, NC = p = 16: every phase state of
// notes/cpm/cpm.md:128-140 for a pulse truncated to two symbols) — 0.2 dB ahead of the 16-state design BASELINE
// configs[2] names, on the SAME 16 matched filters per symbol, so the front end is unchanged.  The algorithm is the one
// cpm_oracle.c defines (conventions of waveforms/viterbi/algorithm.py:57-98: increment Re(rotation * mf) minimised,
// strict '<' / first listed branch on ties, first arg-min, min-normalised metrics, one decision per call from the best
// state); decisions are bit-identical to it.
//
// Per call every state lane rotates its M matched-filter outputs by its survivor's phase, drops the M candidates into
// the LDS slots of the end states they lead to (slot j of an end state = its j-th listed branch: start state ascending,
// then input ascending), reads its own M incoming candidates back, picks the first minimum, fetches the winner's phase
// index and decision register by ds_bpermute, and joins a 64-lane all-reduce for the min-normalisation (DPP inside the
// rows, four v_readlane across them).  Chunk-parallel with the same proof as the other forms: a chunk starts `warmup`
// calls early, records the state its own calls started from and ended with (3 words per state: 384 words per chunk), a
// small kernel compares neighbours bitwise, and a second launch re-runs the chunks that missed — here as a PAIR of waves
// per chunk (one from the state the warm-up arrived at, one from the true state) that compare through LDS after every
// batch until they meet.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "wf_cpm_detect.h"

#define WIDE_TB 4               // calls per staged batch of rows
#define WIDE_XS 68              // exchange: candidate j of end lane L at word j * WIDE_XS + L (8 B words: conflict-free column reads)
#define WIDE_WAVES 4            // detectors (chunks) per workgroup of the first launch
#define WIDE_THREADS (64 * WIDE_WAVES)

struct cpm_wide_params {
    int M, p, nh, K0, K1, Lp, NC, D, S, NF;
    int CH, W;
    int64_t ncalls, nchunks;
    int rows_off, xch_off, dec_off, wave_bytes, rot_off, cmp_off;   // dynamic LDS layout (bytes)
    // variant kv: 0 / 1 = the symbol leaving the window uses K[0] / K[1]; 2 = a virtual pre-start symbol (no phase).
    // dest[kv][s][u] = 4 * end_state + slot of branch (s, u); info[kv][e][j] = src | u << 6 | delta << 8 for slot j of end
    // state e, delta = what the branch adds to the survivor's TILTED phase index r = (2 v - tilt) mod 2p.
    uint8_t dest[3][64][4];
    uint16_t info[3][64][4];
};

__device__ __forceinline__ double wide_min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// all-reduce min over the 64 lanes: row_ror 8, 4, 2, 1 inside each row, then the four row minima through the scalar file
__device__ __forceinline__ double wide_wave_min(double v)
{
    v = wide_min_raw(v, wf_dpp_f64<0x128, 0xf>(v));
    v = wide_min_raw(v, wf_dpp_f64<0x124, 0xf>(v));
    v = wide_min_raw(v, wf_dpp_f64<0x122, 0xf>(v));
    v = wide_min_raw(v, wf_dpp_f64<0x121, 0xf>(v));
    const long long b = __double_as_longlong(v);
    const int lo = (int)b, hi = (int)(b >> 32);
    double q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = __builtin_amdgcn_readlane(lo, 16 * k), h = __builtin_amdgcn_readlane(hi, 16 * k);
        q[k] = __longlong_as_double(((long long)h << 32) | (unsigned)l);
    }
    return wide_min_raw(wide_min_raw(q[0], q[1]), wide_min_raw(q[2], q[3]));      // (no NaNs among metrics: min is exact and order-free; fmin would quiet each bit-cast operand first)
}

__device__ __forceinline__ uint64_t wide_bperm_u64(int byte_addr, uint64_t v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(v >> 32));
    return ((uint64_t)(unsigned)hi << 32) | (unsigned)lo;
}

__device__ __forceinline__ void wide_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Device-resident detector state, wide layout (inside WF_CPM_STATE_BYTES): [0] calls made, [1 + s] metrics,
// [65 + s] tilted phase indices, [129 + s] decision registers; staging copy from word 256.
#define WIDE_ST_N 0
#define WIDE_ST_M 1
#define WIDE_ST_V 65
#define WIDE_ST_H 129
#define WIDE_ST_WORDS 193
#define WIDE_ST_STAGE 256
#define WIDE_EDGE_WORDS (2 * 64 * 3)

template <int M_, int LP_, bool REPAIR>
__device__ __forceinline__ void cpm_wide_body(const double2 *__restrict__ rows, const double *__restrict__ rot,
                                              uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                              uint64_t *__restrict__ edge, unsigned long long *__restrict__ unmerged,
                                              const cpm_wide_params &P, const int64_t chunk, uint64_t *__restrict__ next_count,
                                              uint64_t *__restrict__ next_list)
{
    constexpr int M = M_;
    constexpr int LGM = M_ == 4 ? 2 : 1;
    constexpr int NF = LP_ == 1 ? M_ : (LP_ == 2 ? M_ * M_ : M_ * M_ * M_);
    constexpr int PIECES = WIDE_TB * NF;                      // 16 B pieces per batch
    constexpr int PL = PIECES >= 64 ? PIECES / 64 : 1;        // pieces per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s = lane;
    const bool active = s < P.S;
    const int corr = s / P.NC;
    char *wbase = smem + wave * P.wave_bytes;
    double2 *rowbuf = reinterpret_cast<double2 *>(wbase + P.rows_off);
    double *xch = reinterpret_cast<double *>(wbase + P.xch_off);      // [4 slots][WIDE_XS]
    uint8_t *dec = reinterpret_cast<uint8_t *>(wbase + P.dec_off);

    const int64_t n0 = state ? (int64_t)state[WIDE_ST_N] : 0;          // calls made before this launch
    const int64_t k_first = chunk * P.CH;                              // first own call (local index)
    const bool live = k_first < P.ncalls;
    const int T = P.W + P.CH;

    uint32_t dsel[3], ilo[3], ihi[3];
#pragma unroll
    for (int kv = 0; kv < 3; ++kv) {
        dsel[kv] = *reinterpret_cast<const uint32_t *>(&P.dest[kv][s][0]);
        ilo[kv] = *reinterpret_cast<const uint32_t *>(&P.info[kv][s][0]);
        ihi[kv] = *reinterpret_cast<const uint32_t *>(&P.info[kv][s][2]);
    }

    double m = active ? 0.0 : INFINITY;
    const int64_t k_start = chunk == 0 ? 0 : k_first - P.W;            // first call this wave really runs
    int r = 2 * (s % P.NC) - cpm_tilt(P.M, P.p, P.nh, P.K0, P.K1, P.Lp, n0 + k_start);
    r += r < 0 ? 2 * P.p : 0;
    uint64_t hist = 0;
    if (state && chunk == 0 && n0 > 0) {                               // continue the carried detector
        m = active ? __longlong_as_double((long long)state[WIDE_ST_M + s]) : INFINITY;
        r = (int)state[WIDE_ST_V + s];
        hist = state[WIDE_ST_H + s];
    }
    uint64_t *const erec = edge + chunk * WIDE_EDGE_WORDS;
    if constexpr (REPAIR) {
        const uint64_t *src = wave ? erec - WIDE_EDGE_WORDS + WIDE_EDGE_WORDS / 2 : erec;   // wave 1: the previous chunk's end (as it is now) | wave 0: this chunk's recorded start
        const uint64_t w0 = active ? src[3 * s] : 0ull, w1 = active ? src[3 * s + 1] : 0ull, w2 = active ? src[3 * s + 2] : 0ull;
        m = active ? __longlong_as_double((long long)w0) : INFINITY;
        r = (int)w1;
        hist = w2;
        __syncthreads();                                               // wave 0 has the OLD start before wave 1 replaces it
        if (wave == 1 && active) {                                     // what this run starts from becomes the chunk's recorded start (wf_cpm_detect.h)
            erec[3 * s] = w0;
            erec[3 * s + 1] = w1;
            erec[3 * s + 2] = w2;
        }
    }

    auto fetch = [&](int b, double2 (&dst)[PL]) __attribute__((always_inline)) {
        const int64_t kb = k_first - P.W + (int64_t)b * WIDE_TB;       // local call of the batch's first row
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int q = lane + 64 * i;
            const int qq = q < PIECES ? q : 0;
            int64_t row = kb + qq / NF;
            row = row < 0 ? 0 : (row >= P.ncalls ? P.ncalls - 1 : row);   // never decoded when clamped
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(rows + row * NF + qq % NF));
            dst[i] = make_double2(v.x, v.y);
        }
    };
    auto slot_of = [&](uint32_t dsel_kv, int u) __attribute__((always_inline)) {
        const int dst = (int)((dsel_kv >> (8 * u)) & 0xFFu);
        return active ? (dst & 3) * WIDE_XS + (dst >> 2) : u * WIDE_XS + lane;   // (lanes that hold no state park theirs in their own column)
    };
    int xslot[3][M];
#pragma unroll
    for (int kv = 0; kv < 3; ++kv)
#pragma unroll
        for (int u = 0; u < M; ++u) xslot[kv][u] = slot_of(dsel[kv], u);
    const double2 *zlane = rowbuf + M * corr;
    const int dshift = LGM * (P.D - 1);

    // One detector call.  KV: the leaving symbol's variant as a compile-time constant (0 / 1), or -1 = per call
    // (virtual pre-start symbols, ragged ends).  FAST: a real call that either emits or not — no predication.
    auto step = [&](auto KVc, auto FASTc, int tt, int t, bool emit) __attribute__((always_inline)) {
        constexpr int KV = decltype(KVc)::value;
        constexpr bool FAST = decltype(FASTc)::value;
        const int64_t k = k_first - P.W + t;                            // local call index
        const int64_t n = n0 + k;                                       // global call index
        const bool valid = FAST || (live && k >= 0 && k < P.ncalls);    // (wave-uniform)
        int kv = KV;
        if (KV < 0) {
            const int64_t m_old = n - LP_ + 1;
            kv = m_old < 0 ? 2 : (P.nh == 2 ? (int)(m_old & 1) : 0);
        }
        const uint32_t il = kv == 0 ? ilo[0] : (kv == 1 ? ilo[1] : ilo[2]);
        const uint32_t ih = kv == 0 ? ihi[0] : (kv == 1 ? ihi[1] : ihi[2]);
        const double2 cs = make_double2(rot[r], rot[CPM_ROT_SIN + r]);
        const double2 *zrow = zlane + tt * NF;
#pragma unroll
        for (int u = 0; u < M; ++u) {
            const double2 z = zrow[u];
            const double inc = -fma(cs.x, z.x, cs.y * z.y);             // -Re(e^{-j theta} Z)
            xch[kv == 0 ? xslot[0][u] : (kv == 1 ? xslot[1][u] : xslot[2][u])] = m + inc;
        }
        wide_wave_sync();
        double c[M];
#pragma unroll
        for (int j = 0; j < M; ++j) c[j] = xch[j * WIDE_XS + lane];
        double best;
        uint32_t inf;                                                   // the winner's table entry in the low 16 bits
        if constexpr (M == 4) {
            const bool f01 = c[1] < c[0], f23 = c[3] < c[2];
            const double b01 = wide_min_raw(c[0], c[1]), b23 = wide_min_raw(c[2], c[3]);
            const bool f = b23 < b01;
            best = wide_min_raw(b01, b23);
            inf = (f ? ih : il) >> ((f ? f23 : f01) ? 16 : 0);
        } else {
            const bool f01 = c[1] < c[0];
            best = wide_min_raw(c[0], c[1]);
            inf = il >> (f01 ? 16 : 0);
        }
        const int src = (int)(inf & 63u), u_new = (int)((inf >> 6) & 3u), delta = (int)((inf >> 8) & 0x7Fu);
        const int baddr = src << 2;
        const uint32_t nr_raw = (uint32_t)(__builtin_amdgcn_ds_bpermute(baddr, r) + delta);   // < 4p
        const int nr = (int)min(nr_raw, nr_raw - (uint32_t)(2 * P.p));   // mod 2p
        const uint64_t nh_ = (wide_bperm_u64(baddr, hist) << LGM) | (uint64_t)u_new;
        double nm = best;      // (lanes that hold no state: +inf by construction)
        nm -= wide_wave_min(nm);                                        // the minimum becomes exactly 0.0
        if (valid) {
            m = nm;
            r = nr;
            hist = nh_;
        }
        if (emit && valid) {
            const unsigned long long zero = __builtin_amdgcn_ballot_w64(nm == 0.0);
            if (__builtin_amdgcn_inverse_ballot_w64(zero & (0ull - zero)))   // np.argmin: the first state whose metric is the minimum
                dec[t - P.W] = (n >= P.D - 1) ? (uint8_t)((nh_ >> dshift) & (uint64_t)(M - 1)) : (uint8_t)0;
        }
    };
    using kv0 = std::integral_constant<int, 0>;
    using kv1 = std::integral_constant<int, 1>;
    using kvd = std::integral_constant<int, -1>;
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // parity of the leaving symbol at the first call of a batch (chunk starts, warm-up and batch length are even)
    const int par_u = __builtin_amdgcn_readfirstlane(P.nh == 2 ? (int)((n0 + k_first - P.W - LP_ + 1) & 1) : 0);

    double2 pend0[PL], pend1[PL];
    const int nbatch = T / WIDE_TB;                                     // even: W and CH are multiples of 2 * WIDE_TB
    fetch(0, pend0);
    fetch(1 < nbatch ? 1 : 0, pend1);
    auto batch = [&](int b, double2 (&pend)[PL]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int q = lane + 64 * i;
            if (q < PIECES) rowbuf[q] = pend[i];
        }
        fetch(b + 2 < nbatch ? b + 2 : nbatch - 1, pend);               // issued unconditionally
        wide_wave_sync();
        const int t0 = b * WIDE_TB;
        if (!REPAIR && t0 == P.W && live) {                             // the next call is the chunk's first own one
            erec[3 * s] = (uint64_t)__double_as_longlong(m);
            erec[3 * s + 1] = (uint64_t)(int64_t)r;
            erec[3 * s + 2] = hist;
        }
        const bool emit = t0 >= P.W;
        const int64_t kb = k_first - P.W + t0;
        const bool easy = live && kb >= 0 && kb + WIDE_TB <= P.ncalls && n0 + kb - LP_ + 1 >= 0 && (!emit || n0 + kb >= P.D - 1);
        if (easy) {                                                     // (wave-uniform)
            if (P.nh == 1) {
                step(kv0{}, yes{}, 0, t0, emit); step(kv0{}, yes{}, 1, t0 + 1, emit);
                step(kv0{}, yes{}, 2, t0 + 2, emit); step(kv0{}, yes{}, 3, t0 + 3, emit);
            } else if (par_u == 0) {
                step(kv0{}, yes{}, 0, t0, emit); step(kv1{}, yes{}, 1, t0 + 1, emit);
                step(kv0{}, yes{}, 2, t0 + 2, emit); step(kv1{}, yes{}, 3, t0 + 3, emit);
            } else {
                step(kv1{}, yes{}, 0, t0, emit); step(kv0{}, yes{}, 1, t0 + 1, emit);
                step(kv1{}, yes{}, 2, t0 + 2, emit); step(kv0{}, yes{}, 3, t0 + 3, emit);
            }
        } else {
#pragma unroll 1
            for (int tt = 0; tt < WIDE_TB; ++tt) step(kvd{}, no{}, tt, t0 + tt, emit);
        }
        wide_wave_sync();                                               // batch consumed before the next stash
    };
    if constexpr (REPAIR) {
        // (P.W = 0 in this launch.)  The two waves of the workgroup compare their states through LDS after every batch.
        const uint64_t hmask = LGM * P.D >= 64 ? ~0ull : ((1ull << (LGM * P.D)) - 1ull);
        uint64_t *cmp = reinterpret_cast<uint64_t *>(smem + P.cmp_off);      // [2 waves][3][64]
        int done = 0;
        bool merged = false;
        for (int b = 0; b < nbatch && !merged; ++b) {
            if (b & 1) batch(b, pend1); else batch(b, pend0);
            done = (b + 1) * WIDE_TB;
            uint64_t *mine = cmp + wave * 192, *theirs = cmp + (wave ^ 1) * 192;
            mine[lane] = (uint64_t)__double_as_longlong(m);
            mine[64 + lane] = (uint64_t)(int64_t)r;
            mine[128 + lane] = hist & hmask;
            __syncthreads();
            const bool diff = active && (theirs[lane] != (uint64_t)__double_as_longlong(m) || theirs[64 + lane] != (uint64_t)(int64_t)r ||
                                         theirs[128 + lane] != (hist & hmask));
            merged = __builtin_amdgcn_ballot_w64(diff) == 0ull;         // (symmetric: both waves get the same answer)
            __syncthreads();
        }
        if (wave == 1) {                                                // the new trajectory's decisions up to the meeting point
            for (int q = lane; q < done; q += 64)
                if (k_first + q < P.ncalls) out[k_first + q] = dec[q];
            if (!merged) {                                              // the chunk ENDS in another state than before
                if (active) {
                    erec[WIDE_EDGE_WORDS / 2 + 3 * s] = (uint64_t)__double_as_longlong(m);
                    erec[WIDE_EDGE_WORDS / 2 + 3 * s + 1] = (uint64_t)(int64_t)r;
                    erec[WIDE_EDGE_WORDS / 2 + 3 * s + 2] = hist;
                    if (state && k_first + P.CH >= P.ncalls) {          // ... and it owns the burst's last call: the carry
                        state[WIDE_ST_STAGE + WIDE_ST_M + s] = (uint64_t)__double_as_longlong(m);
                        state[WIDE_ST_STAGE + WIDE_ST_V + s] = (uint64_t)(int64_t)r;
                        state[WIDE_ST_STAGE + WIDE_ST_H + s] = hist;
                    }
                }
                if (lane == 0 && chunk + 1 < P.nchunks)                 // the next chunk's start no longer matches: next round
                    next_list[atomicAdd(reinterpret_cast<unsigned long long *>(next_count), 1ull)] = (uint64_t)(chunk + 1);
            }
            if (lane == 0) {
                atomicAdd(unmerged + 1, 1ull);                          // [1]: chunk repairs run, [2]: ... that handed on
                if (!merged) atomicAdd(unmerged + 2, 1ull);
            }
        }
        return;
    }
    for (int b = 0; b < nbatch; b += 2) {
        batch(b, pend0);
        batch(b + 1, pend1);
    }
    if (live) {
        for (int off = 16 * lane; off < P.CH; off += 1024) {            // decisions: 16 B per lane and 1024 calls
            const int64_t k = k_first + off;
            if (k + 16 <= P.ncalls) {
                *reinterpret_cast<uint4 *>(out + k) = *reinterpret_cast<const uint4 *>(dec + off);
            } else {
                for (int q = 0; q < 16 && k + q < P.ncalls; ++q) out[k + q] = dec[off + q];
            }
        }
        erec[WIDE_EDGE_WORDS / 2 + 3 * s] = (uint64_t)__double_as_longlong(m);      // proof record: what this chunk ended with
        erec[WIDE_EDGE_WORDS / 2 + 3 * s + 1] = (uint64_t)(int64_t)r;
        erec[WIDE_EDGE_WORDS / 2 + 3 * s + 2] = hist;
    }
    if (state && live && k_first + P.CH >= P.ncalls) {                 // the wave that owns the last call
        if (lane == 0) state[WIDE_ST_STAGE + WIDE_ST_N] = (uint64_t)(n0 + P.ncalls);
        state[WIDE_ST_STAGE + WIDE_ST_M + s] = (uint64_t)__double_as_longlong(m);
        state[WIDE_ST_STAGE + WIDE_ST_V + s] = (uint64_t)(int64_t)r;
        state[WIDE_ST_STAGE + WIDE_ST_H + s] = hist;
    }
}

__device__ __forceinline__ double *wide_stage_rot(const double2 *__restrict__ rot_cs, const cpm_wide_params &P)   // the caller synchronises
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *rot = reinterpret_cast<double *>(smem + P.rot_off);       // cos at [r], sin at [CPM_ROT_SIN + r]
    for (int k = threadIdx.x; k < 2 * P.p; k += blockDim.x) {
        const double2 e = rot_cs[k];
        rot[k] = e.x;
        rot[CPM_ROT_SIN + k] = e.y;
    }
    return rot;
}

template <int M_, int LP_>
__global__ __launch_bounds__(WIDE_THREADS) void cpm_wide_kernel(const double2 *__restrict__ rows, const double2 *__restrict__ rot_cs,
                                                               uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                               uint64_t *__restrict__ edge, unsigned long long *__restrict__ unmerged,
                                                               cpm_wide_params P)
{
    const double *rot = wide_stage_rot(rot_cs, P);
    if (blockIdx.x == 0 && threadIdx.x < CPM_NLIST) cpm_list_counts(edge, P.nchunks, WIDE_EDGE_WORDS)[threadIdx.x] = 0;   // the repair lists: empty
    __syncthreads();
    cpm_wide_body<M_, LP_, false>(rows, rot, out, state, edge, unmerged, P, (int64_t)blockIdx.x * WIDE_WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nullptr, nullptr);
}

// One repair round (list layout and invariant: wf_cpm_detect.h): a workgroup of TWO waves per listed chunk — wave 0 from the
// chunk's recorded start, wave 1 from the previous chunk's end — comparing through LDS after every batch.  finisher != 0:
// one workgroup that goes on, round after round, until a round hands nothing on.
template <int M_, int LP_>
__global__ __launch_bounds__(128) void cpm_wide_repair_kernel(const double2 *__restrict__ rows, const double2 *__restrict__ rot_cs,
                                                              uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                              uint64_t *__restrict__ edge, unsigned long long *__restrict__ unmerged,
                                                              cpm_wide_params P, int lin, int lout, int finisher)
{
    uint64_t *const counts = cpm_list_counts(edge, P.nchunks, WIDE_EDGE_WORDS);
    int64_t n = (int64_t)__hip_atomic_load(&counts[lin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == 0) return;                                               // (the whole grid)
    const double *rot = wide_stage_rot(rot_cs, P);
    __syncthreads();
    for (;;) {
        const uint64_t *list = cpm_list(edge, P.nchunks, WIDE_EDGE_WORDS, lin);
        for (int64_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
            const uint64_t cw = __hip_atomic_load(&list[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int64_t chunk = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(cw >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)cw));   // (uniform, and said so)
            cpm_wide_body<M_, LP_, true>(rows, rot, out, state, edge, unmerged, P, chunk, &counts[lout], cpm_list(edge, P.nchunks, WIDE_EDGE_WORDS, lout));
            __syncthreads();                                          // (the compare area and the waves' buffers are reused)
        }
        if (!finisher) return;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&counts[lin], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: the next round's output
        n = (int64_t)__hip_atomic_load(&counts[lout], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __syncthreads();
        if (n == 0) return;
        const int t = lin;
        lin = lout;
        lout = t;
    }
}

// Every chunk against its predecessor: thread = (chunk c >= 1, state s); failed chunks are LISTED behind the records
// for the repair launches, or (repair = 0) counted as unproven.
__global__ void cpm_wide_verify_kernel(uint64_t *__restrict__ edge, int64_t nchunks, int S, uint64_t hmask,
                                       unsigned long long *__restrict__ unmerged, int repair)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c = idx / 64 + 1;
    const int s = (int)(idx & 63);
    bool bad = false;
    if (c < nchunks && s < S) {
        const uint64_t *a = edge + c * WIDE_EDGE_WORDS, *b = edge + (c - 1) * WIDE_EDGE_WORDS + WIDE_EDGE_WORDS / 2;
        bad = a[3 * s] != b[3 * s] || a[3 * s + 1] != b[3 * s + 1] || ((a[3 * s + 2] ^ b[3 * s + 2]) & hmask) != 0ull;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(bad);
    if (s == 0 && m) {
        if (repair) {
            unsigned long long *counts = reinterpret_cast<unsigned long long *>(cpm_list_counts(edge, nchunks, WIDE_EDGE_WORDS));
            cpm_list(edge, nchunks, WIDE_EDGE_WORDS, 0)[atomicAdd(counts, 1ull)] = (uint64_t)c;
        } else {
            atomicAdd(unmerged, 1ull);
        }
    }
}

__global__ void cpm_wide_commit_kernel(uint64_t *state)
{
    for (int t = threadIdx.x; t < WIDE_ST_WORDS; t += blockDim.x) state[t] = state[WIDE_ST_STAGE + t];
}

// Host: the trellis permutation, enumerated exactly like the sequential statement (start state ascending, then input
// ascending), so slot j of an end state is the j-th listed branch into it and strict '<' over slots 0, 1, ... is its tie-break.
static int wide_build_tables(const wf_cpm_detector_config *d, cpm_wide_params &P)
{
    const int M = d->M, Lp = d->Lp, NC = d->NC, p = d->p;
    const int lgM = M == 4 ? 2 : 1;
    int ncorr = 1, msub = 1, NF = 1;
    for (int i = 1; i < Lp; ++i) ncorr *= M;
    for (int i = 2; i < Lp; ++i) msub *= M;
    for (int i = 0; i < Lp; ++i) NF *= M;
    const int S = NC * ncorr;
    WF_REQUIRE(S > 16 && S <= 64, "wf_cpm (wide form): %d states", S);
    WF_REQUIRE(d->D * lgM <= 64, "wf_cpm: decision delay %d does not fit the 64-bit decision register", d->D);
    P.M = M; P.p = p; P.nh = d->nh; P.K0 = d->K[0]; P.K1 = d->nh == 2 ? d->K[1] : d->K[0];
    P.Lp = Lp; P.NC = NC; P.D = d->D; P.S = S; P.NF = NF;
    memset(P.dest, 0xFF, sizeof P.dest);
    memset(P.info, 0xFF, sizeof P.info);
    for (int kv = 0; kv < 3; ++kv) {
        const int K_old = kv == 2 ? 0 : (kv == 1 ? P.K1 : P.K0);
        int fill[64] = {0};
        for (int s = 0; s < S; ++s) {
            const int cls = s % NC, corr = s / NC;
            for (int u = 0; u < M; ++u) {
                const int u_old = Lp == 1 ? u : corr / msub;
                const int corr2 = Lp == 1 ? 0 : u + M * (corr % msub);
                const int inc = (K_old * u_old) % p;
                const int s2 = (cls + inc) % NC + NC * corr2;
                const int j = fill[s2]++;
                WF_REQUIRE(j < M, "wf_cpm: internal: more than M branches into a state");
                P.dest[kv][s][u] = (uint8_t)(4 * s2 + j);
                const int delta = ((2 * inc - (M - 1) * K_old) % (2 * p) + 2 * p) % (2 * p);
                P.info[kv][s2][j] = (uint16_t)(s | (u << 6) | (delta << 8));
            }
        }
        for (int s = 0; s < S; ++s) WF_REQUIRE(fill[s] == M, "wf_cpm: internal: state %d has %d incoming branches", s, fill[s]);
    }
    return WF_OK;
}

int wf_cpm_wide_applies(const wf_cpm_detector_config *d)
{
    if (!d || !(d->M == 2 || d->M == 4) || d->Lp < 1 || d->Lp > 3 || d->NC < 1) return 0;
    int ncorr = 1;
    for (int i = 1; i < d->Lp; ++i) ncorr *= d->M;
    const int S = d->NC * ncorr;
    return S > 16 && S <= 64;
}

// Chunk length of the wide form: the smallest multiple of 64 that puts the burst into one round of resident waves (8
// workgroups of 4 detectors per CU), at least 256 and 2 W (chunk_opt: WF_OPT_CPM_CHUNK_CALLS, 0 = this rule).
int64_t wf_cpm_wide_chunk_calls(int64_t ncalls, int W, int cus, int64_t chunk_opt)
{
    const int64_t slots = (int64_t)cus * 8 * WIDE_WAVES;
    int64_t ch = ((ncalls + slots - 1) / slots + 63) / 64 * 64;
    if (ch < 256) ch = 256;
    if (ch < 2 * W) ch = (2 * W + 63) / 64 * 64;
    if (chunk_opt > 0) ch = (chunk_opt + 63) / 64 * 64;
    if (ch > 8192) ch = 8192;
    return ch;
}

// Default warm-up: 64 calls (chunks that miss it are repaired by the launches behind the first); a multiple of two batches.
// 1e7 calls are 7813 chunks of 1280, so the warm-up is a tenth of the work: ARTM 64 states, same box, steady state per block —
// 160 calls 3.02 - 3.03 ms (0.15 repairs per block at 0 dB); 96: 2.86 - 2.88 (33 at 0 dB); 64: 2.78 at 10 dB, 2.83 at 6 dB
// (55 repairs per block), 2.85 at 0 dB (450); 48: 2.76 / 2.80 / 2.88 (1544) — profiles/r06_ab_big_trellis_warmup.log.
int wf_cpm_wide_warmup(int warmup)
{
    int W = warmup ? warmup : 64;
    W = (W + 2 * WIDE_TB - 1) / (2 * WIDE_TB) * (2 * WIDE_TB);
    return W > 4096 ? 4096 : W;
}

int wf_cpm_wide_detect(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri, int64_t ncalls,
                       int warmup, uint8_t *d_decisions, void *d_state, void *stream)
{
    cpm_wide_params P{};
    WF_REQUIRE((det->M == 2 || det->M == 4) && det->Lp >= 1 && det->Lp <= 3 && (det->nh == 1 || det->nh == 2) && det->p >= 1 && det->p <= 64 &&
                   det->NC >= 1 && det->p % det->NC == 0 && det->D >= 1,
               "wf_cpm: unsupported detector (M %d Lp %d nh %d p %d NC %d D %d)", det->M, det->Lp, det->nh, det->p, det->NC, det->D);
    for (int i = 0; i < det->nh; ++i) WF_REQUIRE(det->K[i] >= 0 && det->K[i] < det->p, "wf_cpm: K[%d] = %d outside [0, p)", i, det->K[i]);
    int rc = wide_build_tables(det, P);
    if (rc) return rc;
    const int W = wf_cpm_wide_warmup(warmup);
    P.CH = (int)wf_cpm_wide_chunk_calls(ncalls, W, ctx->cus, ctx->opt[WF_OPT_CPM_CHUNK_CALLS]);
    P.W = W;
    P.ncalls = ncalls;
    const int pieces = WIDE_TB * P.NF;
    P.rows_off = 0;
    P.xch_off = pieces * 16;
    P.dec_off = P.xch_off + 4 * WIDE_XS * 8;
    P.wave_bytes = (P.dec_off + P.CH + 15) / 16 * 16;
    P.rot_off = WIDE_WAVES * P.wave_bytes;
    P.cmp_off = P.rot_off + 2 * CPM_ROT_SIN * 8;
    const size_t lds = (size_t)P.cmp_off + 2 * 192 * 8;
    WF_REQUIRE(lds <= 160 * 1024, "wf_cpm_viterbi_detect: chunk of %d calls does not fit LDS", P.CH);
    const int64_t nchunks = (ncalls + P.CH - 1) / P.CH;
    const int64_t nblocks = (nchunks + WIDE_WAVES - 1) / WIDE_WAVES;
    WF_REQUIRE(nblocks < (1ll << 31), "wf_cpm_viterbi_detect: burst too long for one launch");
    P.nchunks = nchunks;
    rc = wf_ctx_reserve_vit(ctx, cpm_edge_total_words(nchunks, WIDE_EDGE_WORDS));
    if (rc) return rc;
    uint64_t *edge = reinterpret_cast<uint64_t *>(ctx->d_vit_edge);
    hipStream_t s = wf_stream(stream);
    using kern_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, unsigned long long *, cpm_wide_params);
    using repair_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, unsigned long long *, cpm_wide_params, int, int, int);
    kern_t k = nullptr;
    repair_t kr = nullptr;
    if (P.M == 4) {
        k = P.Lp == 1 ? cpm_wide_kernel<4, 1> : (P.Lp == 2 ? cpm_wide_kernel<4, 2> : cpm_wide_kernel<4, 3>);
        kr = P.Lp == 1 ? cpm_wide_repair_kernel<4, 1> : (P.Lp == 2 ? cpm_wide_repair_kernel<4, 2> : cpm_wide_repair_kernel<4, 3>);
    } else {
        k = P.Lp == 1 ? cpm_wide_kernel<2, 1> : (P.Lp == 2 ? cpm_wide_kernel<2, 2> : cpm_wide_kernel<2, 3>);
        kr = P.Lp == 1 ? cpm_wide_repair_kernel<2, 1> : (P.Lp == 2 ? cpm_wide_repair_kernel<2, 2> : cpm_wide_repair_kernel<2, 3>);
    }
    if (lds > 48 * 1024) {
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(WIDE_THREADS), lds, s, reinterpret_cast<const double2 *>(d_rows_ri),
                       reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), edge, ctx->d_vit_unmerged, P);
    WF_LAUNCH_CHECK();
    if (nchunks > 1) {
        const int lgM = P.M == 4 ? 2 : 1;
        const uint64_t hmask = lgM * P.D >= 64 ? ~0ull : ((1ull << (lgM * P.D)) - 1ull);
        const int repair = ctx->opt[WF_OPT_DET_REPAIR] == 0 ? 1 : 0;
        const dim3 vgrid((unsigned)(((nchunks - 1) * 64 + 255) / 256));
        hipLaunchKernelGGL(cpm_wide_verify_kernel, vgrid, dim3(256), 0, s, edge, nchunks, P.S, hmask, ctx->d_vit_unmerged, repair);
        WF_LAUNCH_CHECK();
        if (repair) {
            cpm_wide_params Pr = P;
            Pr.W = 0;
            for (int round = 0; round < 3; ++round) {               // two parallel rounds, then the finisher (wf_cpm_detect.h)
                hipLaunchKernelGGL(kr, dim3(round < 2 ? 4 * CPM_REPAIR_BLOCKS : 1), dim3(128), lds, s, reinterpret_cast<const double2 *>(d_rows_ri),
                                   reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), edge,
                                   ctx->d_vit_unmerged, Pr, round, round + 1, round == 2 ? 1 : 0);
                WF_LAUNCH_CHECK();
            }
            if (ctx->opt[WF_OPT_DET_FINAL_VERIFY]) {
                hipLaunchKernelGGL(cpm_wide_verify_kernel, vgrid, dim3(256), 0, s, edge, nchunks, P.S, hmask, ctx->d_vit_unmerged, 0);
                WF_LAUNCH_CHECK();
            }
        }
    }
    if (d_state) {
        hipLaunchKernelGGL(cpm_wide_commit_kernel, dim3(1), dim3(256), 0, s, static_cast<uint64_t *>(d_state));
        WF_LAUNCH_CHECK();
    }
    return WF_OK;
}
