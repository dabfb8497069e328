// wf_cpm_detect.hip — generic CPM trellis detector (ARTM multi-h CPM, PCM/FM): matched-filter
// rows, chunk-parallel Viterbi, error count, and the device-resident link built from them.
//
// mcdiarmid/waveforms has no detector for these waveforms (only their modulators:
// waveforms/cpm/multih/*, waveforms/cpm/pcmfm/*, and the theory of notes/cpm/cpm.md:52-140), so
// the algorithm is the one DEFINED by cpm_oracle.c (the sequential statement kept with the tests), which follows the conventions of the
// reference's one detector (waveforms/viterbi/algorithm.py:57-98: increment Re(rotation * mf)
// minimised, strict '<' / first listed branch on ties, first arg-min, min-normalised metrics, one
// decision per call from the best state) on the tilted-phase trellis of notes/cpm/cpm.md:100-140
// with pulse-truncation matched filters in the manner of examples/soqpsk_detection.py:134-156.
// The kernels here reproduce that sequential detector bit for bit.
//
// Mapping.  A trellis of <= 16 states lives in one DPP ROW: lane = state, 16 lanes = one detector,
// 4 detectors (chunks of consecutive calls) per wave.  Per call every state lane
//   - rotates its M matched-filter outputs by its survivor's phase (table in LDS), forms the M
//     candidate metrics and drops each into the slot of the end state it leads to (LDS exchange:
//     the trellis permutation is data-independent, slots are precomputed on the host),
//   - reads back its own M incoming candidates, selects with strict '<' in list order, fetches the
//     winner's phase index and decision register with ds_bpermute,
//   - joins a 16-lane all-reduce (DPP row_ror) for the min-normalisation; the lane whose
//     normalised metric is exactly 0.0 (first such lane) is the best state and emits the decision.
// Chunks start `warmup` calls early from a fresh detector and every launch PROVES that each chunk
// began from bitwise the (metric, phase index, decision register) its predecessor ended with, as
// wf_viterbi.hip does for the 4-state SOQPSK detector.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "wf_cpm_detect.h"

#define CPM_THREADS 256
#define CPM_WAVES (CPM_THREADS / WF_WAVE)
#define CPM_GROUPS 4            // 16-lane detectors per wave
#define CPM_TB 4                // calls per staged batch of rows
#define CPM_DEFAULT_WARMUP 256
// Candidate exchange of one wave: SLOT-major, candidate j of end lane L at word j * CPM_XS + L.  A lane's M
// incoming candidates are then M conflict-free 8 B column reads (two ds_read2_b64), and the scattered
// writes of one input symbol u — in every group the 4 (or NC) end states with that newest symbol, slots
// 0..M-1 — fall into distinct banks: 8 B words, stride 68 = 64 + 4, so slot j starts 8 banks further.
// (The first layout, [lane][4 slots], put the 16-lane passes of the 16 B reads on 8 bank groups — two-way
// conflicts — and all four groups' writes of one symbol on the same 32 banks: 16 % of the LDS cycles of the
// ARTM detector and 42 % of the PCM/FM one were bank conflicts, profiles/r03_pmc_cpmvit_*.json.)
#define CPM_XS 68
// (rotation table: cos at [r], sin at [CPM_ROT_SIN + r], r < 2p <= 128 — a compile-time distance: one ds_read2_b64)

struct cpm_tables {
    // variant kv: 0 / 1 = the symbol leaving the window uses K[0] / K[1]; 2 = it is a virtual
    // pre-start symbol (no phase).  dest[kv][s][u] = 4 * end_state + slot of branch (s, u);
    // info[kv][e][j] = src | u << 4 | delta << 8 for slot j of end state e, delta = (2 (K u_old mod p) - (M - 1) K) mod 2p:
    // what the branch adds to the survivor's TILTED phase index r = (2 v - tilt) mod 2p (v: phase index mod p,
    // tilt: (M - 1) * sum of K over the symbols that left the window) — the rotation table's own index.
    uint8_t dest[3][16][4];
    uint16_t info[3][16][4];
};

struct cpm_vit_params {
    int M, lgM, p, nh, K0, K1, Lp, NC, D, S, NF;
    int CH, W;
    int64_t ncalls, nchunks;
    int rows_off, xch_off, dec_off, wave_bytes, rot_off;   // dynamic LDS layout (bytes)
    // repairs of a launch whose lanes ran the matched filters themselves (cpm_mf_source): `rows` are the noisy samples
    const double *mf_templ;
    int64_t mf_nsamp;
    int mf_col0;
    cpm_tables T;
};

__device__ __forceinline__ int cpm_tilt_at(const cpm_vit_params &P, int64_t n)   // (M-1) * sum K over symbols 0 .. n-Lp, mod 2p
{
    return cpm_tilt(P.M, P.p, P.nh, P.K0, P.K1, P.Lp, n);
}

// all-reduce min over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1)
// (v_min_f64 directly: fmin() first canonicalises operands that arrive through a bit cast — one
// v_max_f64 per operand — and there are no NaNs to quieten here)
__device__ __forceinline__ double cpm_min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ double cpm_row_min(double v)
{
    v = cpm_min_raw(v, wf_dpp_f64<0x128, 0xf>(v));
    v = cpm_min_raw(v, wf_dpp_f64<0x124, 0xf>(v));
    v = cpm_min_raw(v, wf_dpp_f64<0x122, 0xf>(v));
    v = cpm_min_raw(v, wf_dpp_f64<0x121, 0xf>(v));
    return v;
}

__device__ __forceinline__ double cpm_bperm_f64(int byte_addr, double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)b);
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ uint64_t cpm_bperm_u64(int byte_addr, uint64_t v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)v);
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(v >> 32));
    return ((uint64_t)(unsigned)hi << 32) | (unsigned)lo;
}

// 16 B load as a VALUE (a double2 class assignment into a private array lowers to a memcpy and
// keeps the array in scratch)
__device__ __forceinline__ double2 vit_ld16_c(const double2 *p)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = *reinterpret_cast<const v2d *>(p);
    return make_double2(v.x, v.y);
}

__device__ __forceinline__ void cpm_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (device-resident detector state and the per-chunk proof records: wf_cpm_detect.h)

// Waves per SIMD the register allocator is held to: 5 (<= 96 registers) for the trellises with up to 4 filters
// per call (binary with a pulse of <= 2 symbols, any alphabet with one-symbol filters: 88 .. 93 registers; left
// alone they took 97 / 98 / 101), 4 (<= 128) for ARTM's 16 filters per call, 2 for the 64-filter design.
#ifndef CPM_MIN_WAVES
#define CPM_MIN_WAVES(M, LP) (((M) == 2 && (LP) <= 2) || (LP) == 1 ? 5 : ((M) == 4 && (LP) == 3 ? 2 : 4))
#endif
// REPAIR (cpm_repair_kernel) = the launches behind the first one of a detector call (see cpm_verify_kernel and the list
// layout in wf_cpm_detect.h): one WAVE per listed chunk.  The chunk's own calls are run again from BOTH states — groups
// 0 / 2 from the state its record says it started from (what the previous run of this chunk began with: its warm-up's
// arrival, or an earlier repair's input), groups 1 / 3 from the state the previous chunk ends with NOW, which becomes
// the chunk's recorded start — until the two are bitwise equal: from there on the previous run's decisions and end state
// stand, and up to there the second trajectory's decisions replace them.  A pair that has not met by the end of the
// chunk means the chunk's END changed: it is recorded, and the next chunk is listed for the following round (its start
// no longer equals its predecessor's end).  Most chunks that miss a SHORT warm-up meet within a few dozen calls, so the
// warm-up can be sized for the typical merge depth instead of its 1e-7 tail — and is a matter of speed only.
template <int M_, int LP_, bool REPAIR>
__device__ __forceinline__ void cpm_viterbi_body(const double2 *__restrict__ rows, const double *__restrict__ rot,
                                                 uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                 uint64_t *__restrict__ edge, unsigned long long *__restrict__ unmerged,
                                                 const cpm_vit_params &P, const int64_t chunk, uint64_t *__restrict__ next_count,
                                                 uint64_t *__restrict__ next_list)
{
    constexpr int M = M_;
    constexpr int LGM = M_ == 4 ? 2 : 1;
    constexpr int NF = LP_ == 1 ? M_ : (LP_ == 2 ? M_ * M_ : M_ * M_ * M_);
    constexpr int PIECES = CPM_TB * NF;                       // 16 B pieces per group and batch
    constexpr int PL = PIECES >= 16 ? PIECES / 16 : 1;        // pieces per lane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, s = lane & 15;
    const bool active = s < P.S;
    const int corr = s / P.NC;
    char *wbase = smem + wave * P.wave_bytes;
    double2 *rowbuf = reinterpret_cast<double2 *>(wbase + P.rows_off) + g * PIECES;
    double *xch = reinterpret_cast<double *>(wbase + P.xch_off);      // [4 slots][CPM_XS] (see CPM_XS)
    uint8_t *dec = reinterpret_cast<uint8_t *>(wbase + P.dec_off) + g * P.CH;

    const int64_t n0 = state ? (int64_t)state[CPM_ST_N] : 0;          // calls made before this launch
    const int64_t k_first = chunk * P.CH;                             // first own call (local index)
    const bool live = k_first < P.ncalls;
    const int T = P.W + P.CH;

    // per-lane slices of the branch tables (three variants of the leaving symbol's K)
    uint32_t dsel[3], ilo[3], ihi[3];
#pragma unroll
    for (int kv = 0; kv < 3; ++kv) {
        dsel[kv] = *reinterpret_cast<const uint32_t *>(&P.T.dest[kv][s][0]);
        ilo[kv] = *reinterpret_cast<const uint32_t *>(&P.T.info[kv][s][0]);
        ihi[kv] = *reinterpret_cast<const uint32_t *>(&P.T.info[kv][s][2]);
    }

    // detector registers of this state
    double m = active ? 0.0 : INFINITY;
    // The survivor's phase is carried as the TILTED index r = (2 v - tilt) mod 2p, the index of its rotation:
    // a branch adds one table constant to it (cpm_tables), where v and the tilt kept apart cost two modular
    // updates and the index arithmetic on every call.
    const int64_t k_start = chunk == 0 ? 0 : k_first - P.W;            // first call this group really runs
    int r = 2 * (s % P.NC) - cpm_tilt_at(P, n0 + k_start);
    r += r < 0 ? 2 * P.p : 0;
    uint64_t hist = 0;
    if (state && chunk == 0 && n0 > 0) {                               // continue the carried detector
        m = active ? __longlong_as_double((long long)state[CPM_ST_M + s]) : INFINITY;
        r = (int)state[CPM_ST_V + s];
        hist = state[CPM_ST_H + s];
    }
    uint64_t *const erec = edge + chunk * CPM_EDGE_WORDS;               // (written only when the chunk is live)
    if constexpr (REPAIR) {
        const uint64_t *src = (g & 1) ? erec - CPM_EDGE_WORDS + 48 : erec;   // the previous chunk's end (as it is now) | this chunk's recorded start
        const uint64_t w0 = active ? src[3 * s] : 0ull, w1 = active ? src[3 * s + 1] : 0ull, w2 = active ? src[3 * s + 2] : 0ull;
        m = active ? __longlong_as_double((long long)w0) : INFINITY;
        r = (int)w1;
        hist = w2;
        // what this run starts from becomes the chunk's recorded start — exactly the words read (another wave may be
        // rewriting the predecessor's end in this very round: then this chunk is listed again, and the record says
        // truthfully what its decisions were computed from)
        if (g == 1 && active) {
            erec[3 * s] = w0;
            erec[3 * s + 1] = w1;
            erec[3 * s + 2] = w2;
        }
    }

    // cooperative row fetch: piece q = s + 16 i of the batch's CPM_TB * NF pieces
    auto fetch = [&](int b, double2 (&dst)[PL]) __attribute__((always_inline)) {
        const int64_t kb = k_first - P.W + (int64_t)b * CPM_TB;        // local call of the batch's first row
        if constexpr (REPAIR && NF == 16) {
            // The launch's lanes ran the matched filters themselves: no rows exist.  Lane s forms filter s of call kb + i from
            // the samples, as one half of its conjugate pair p = min(s, 15 - s) — the four real chains of cpm_lane_kernel's MF
            // form, operation for operation, so a repaired chunk ends in bitwise the state its successor started from.
            if (P.mf_templ) {
                const double2 *smp = rows;
                const int p = s < 8 ? s : 15 - s;
#pragma unroll
                for (int i = 0; i < PL; ++i) {
                    int64_t row = kb + i;
                    row = row < 0 ? 0 : (row >= P.ncalls ? P.ncalls - 1 : row);
                    const double2 *tp = reinterpret_cast<const double2 *>(P.mf_templ) + ((P.nh == 2 ? (int)((row + P.mf_col0) & 1) : 0) * NF + p) * 9;
                    double sP = 0.0, sQ = 0.0, sR = 0.0, sU = 0.0;
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        int64_t idx = 8 * row + k;
                        idx = idx < P.mf_nsamp ? idx : P.mf_nsamp - 1;
                        const double2 x = smp[idx], tk = tp[k];
                        sP = fma(x.x, tk.x, sP);
                        sU = fma(x.x, -tk.y, sU);
                        sQ = fma(x.y, tk.y, sQ);
                        sR = fma(x.y, tk.x, sR);
                    }
                    dst[i] = s < 8 ? make_double2(sP + sQ, sU + sR) : make_double2(sP - sQ, -(sU - sR));
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int q = s + 16 * i;
            int64_t row = kb + q / NF;
            row = row < 0 ? 0 : (row >= P.ncalls ? P.ncalls - 1 : row);   // never decoded when clamped
            const int qq = q < PIECES ? q : 0;
            {
                typedef double v2d __attribute__((ext_vector_type(2)));
                const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(rows + row * NF + (qq - (qq / NF) * NF)));
                dst[i] = make_double2(v.x, v.y);
            }
        }
    };
    // LDS word index of the slot each of this lane's M candidates goes to (dest = 4 * end state + slot), for
    // the two real variants; the third (virtual pre-start symbols: the first Lp - 1 calls of a burst) is
    // worked out where it is used.  Lanes that hold no state park theirs in their own column, which only
    // they read (and ignore).
    auto slot_of = [&](uint32_t dsel_kv, int u) __attribute__((always_inline)) {
        const int dst = (int)((dsel_kv >> (8 * u)) & 0xFFu);
        return active ? (dst & 3) * CPM_XS + g * 16 + (dst >> 2) : u * CPM_XS + lane;
    };
    int xslot[2][M];
#pragma unroll
    for (int kv = 0; kv < 2; ++kv)
#pragma unroll
        for (int u = 0; u < M; ++u) xslot[kv][u] = slot_of(dsel[kv], u);
    const double2 *zlane = rowbuf + M * corr;
    const int dshift = LGM * (P.D - 1);

    // One detector call.  KV: the leaving symbol's variant as a compile-time constant (0 / 1), or -1
    // = per lane (virtual pre-start symbols, ragged ends).  FAST: every group of the wave runs a
    // real call and all of them either emit or not — no per-lane predication.
    auto step = [&](auto KVc, auto FASTc, int tt, int t, bool emit) __attribute__((always_inline)) {
        constexpr int KV = decltype(KVc)::value;
        constexpr bool FAST = decltype(FASTc)::value;
        const int64_t k = k_first - P.W + t;                            // local call index of this group
        const int64_t n = n0 + k;                                       // global call index
        const bool valid = FAST || (live && k >= 0 && k < P.ncalls);
        int kv = KV;
        if (KV < 0) {
            const int64_t m_old = n - LP_ + 1;
            kv = m_old < 0 ? 2 : (P.nh == 2 ? (int)(m_old & 1) : 0);
        }
        const uint32_t il = kv == 0 ? ilo[0] : (kv == 1 ? ilo[1] : ilo[2]);
        const uint32_t ih = kv == 0 ? ihi[0] : (kv == 1 ? ihi[1] : ihi[2]);
        const double2 cs = make_double2(rot[r], rot[CPM_ROT_SIN + r]);        // (one ds_read2_b64)
        const double2 *zrow = zlane + tt * NF;
        // all M filter outputs first, then the candidates: rows and exchange slots live in the same LDS array,
        // so a read placed after a slot write is kept behind it — read, wait, write, read, wait, ... put M
        // dependent LDS round trips on every call's critical path
        double2 zz[M];
#ifndef CPM_HOIST_Z
#define CPM_HOIST_Z(M) ((M) == 2)   // (M = 4: the 16 extra live registers spill; measured 3 % slower)
#endif
        if (CPM_HOIST_Z(M)) {
#pragma unroll
            for (int u = 0; u < M; ++u) zz[u] = zrow[u];
        }
#pragma unroll
        for (int u = 0; u < M; ++u) {
            const double2 z = CPM_HOIST_Z(M) ? zz[u] : zrow[u];
            const double inc = -fma(cs.x, z.x, cs.y * z.y);             // -Re(e^{-j theta} Z)
            xch[kv == 0 ? xslot[0][u] : (kv == 1 ? xslot[1][u] : slot_of(dsel[2], u))] = m + inc;
        }
        cpm_wave_sync();
        double c[M];
#pragma unroll
        for (int j = 0; j < M; ++j) c[j] = xch[j * CPM_XS + lane];
        // First arg-min over the M slots in list order (strict '<': the first listed branch keeps a tie), as a
        // tree of v_min_f64 + compare — the winner's index only exists as lane masks (the compares' SGPR
        // pairs), combined on the scalar unit; a select chain cost two v_cndmask per comparison more.
        double best;
        uint32_t inf;                                                   // the winner's table entry in the low 16 bits (junk above)
        if constexpr (M == 4) {
            const unsigned long long f01 = __builtin_amdgcn_ballot_w64(c[1] < c[0]), f23 = __builtin_amdgcn_ballot_w64(c[3] < c[2]);
            const double b01 = cpm_min_raw(c[0], c[1]), b23 = cpm_min_raw(c[2], c[3]);
            const unsigned long long f = __builtin_amdgcn_ballot_w64(b23 < b01);
            best = cpm_min_raw(b01, b23);
            const uint32_t pair = __builtin_amdgcn_inverse_ballot_w64(f) ? ih : il;
            inf = pair >> (__builtin_amdgcn_inverse_ballot_w64((f & f23) | (~f & f01)) ? 16 : 0);
        } else {
            const bool f01 = c[1] < c[0];
            best = cpm_min_raw(c[0], c[1]);
            inf = il >> (f01 ? 16 : 0);
        }
        const int src = (int)(inf & 15u), u_new = (int)((inf >> 4) & 3u), delta = (int)((inf >> 8) & 0x7Fu);
        const int baddr = ((lane & 48) | src) << 2;
        const uint32_t nr_raw = (uint32_t)(__builtin_amdgcn_ds_bpermute(baddr, r) + delta);   // < 4p
        const int nr = (int)min(nr_raw, nr_raw - (uint32_t)(2 * P.p));   // mod 2p: the difference wraps to a huge value when nr_raw < 2p
        const uint64_t nh_ = (cpm_bperm_u64(baddr, hist) << LGM) | (uint64_t)u_new;
        double nm = best;      // (lanes that hold no state: metric +inf, candidates parked in their own column, so best = +inf by itself)
        nm -= cpm_row_min(nm);                                          // the minimum becomes exactly 0.0
        if (valid) {
            m = nm;
            r = nr;
            hist = nh_;
        }
        if (emit && (FAST || valid)) {
            // np.argmin: the first state whose metric is the minimum — lowest set bit of each
            // 16-lane field of the ballot, isolated on the scalar unit
            const unsigned long long zero = __builtin_amdgcn_ballot_w64(nm == 0.0);
            unsigned long long firstm = 0;
#pragma unroll
            for (int q = 0; q < CPM_GROUPS; ++q) {
                const unsigned long long f = (zero >> (16 * q)) & 0xFFFFull;
                firstm |= (f & (0ull - f)) << (16 * q);
            }
            if (__builtin_amdgcn_inverse_ballot_w64(firstm))           // (the mask goes straight into exec)
                dec[t - P.W] = (FAST || n >= P.D - 1) ? (uint8_t)((nh_ >> dshift) & (uint64_t)(M - 1)) : (uint8_t)0;
        }
    };
    using kv0 = std::integral_constant<int, 0>;
    using kv1 = std::integral_constant<int, 1>;
    using kvd = std::integral_constant<int, -1>;
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // parity of the leaving symbol at the first call of a batch: the same for every group of the
    // wave and every batch (chunk starts, warm-up and batch length are even)
    const int par0 = P.nh == 2 ? (int)((n0 + k_first - P.W - LP_ + 1) & 1) : 0;
    const int par_u = __builtin_amdgcn_readfirstlane(par0);

    // Two batches of rows in flight per lane: one batch is only ~1 us of detector work, less than a
    // loaded HBM round trip — with a single batch of lookahead every batch waited for memory
    // (1.04 ms for 1e7 calls; the instruction count did not matter).
    double2 pend0[PL], pend1[PL];
    const int nbatch = T / CPM_TB;                                      // even: W and CH are multiples of 2 * CPM_TB
    fetch(0, pend0);
    fetch(1 < nbatch ? 1 : 0, pend1);
    auto batch = [&](int b, double2 (&pend)[PL]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int q = s + 16 * i;
            if (q < PIECES) rowbuf[q] = pend[i];
        }
        fetch(b + 2 < nbatch ? b + 2 : nbatch - 1, pend);               // issued unconditionally (see wf_viterbi.hip)
        cpm_wave_sync();
        const int t0 = b * CPM_TB;
        if (!REPAIR && t0 == P.W && live) {                             // the next call is the chunk's first own one
            erec[3 * s] = (uint64_t)__double_as_longlong(m);
            erec[3 * s + 1] = (uint64_t)(int64_t)r;
            erec[3 * s + 2] = hist;
        }
        const bool emit = t0 >= P.W;
        const int64_t kb = k_first - P.W + t0;
        const bool easy = live && kb >= 0 && kb + CPM_TB <= P.ncalls && n0 + kb - LP_ + 1 >= 0 && (!emit || n0 + kb >= P.D - 1);
        if (__builtin_amdgcn_ballot_w64(easy) == ~0ull) {               // wave-uniform: almost every batch
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
            for (int tt = 0; tt < CPM_TB; ++tt) step(kvd{}, no{}, tt, t0 + tt, emit);
        }
        cpm_wave_sync();                                                // batch consumed before the next stash
    };
    if constexpr (REPAIR) {
        // (P.W = 0 in this launch: the first batch is the chunk's first own call)
        const uint64_t hmask = LGM * P.D >= 64 ? ~0ull : ((1ull << (LGM * P.D)) - 1ull);
        auto met = [&]() __attribute__((always_inline)) {               // both trajectories in bitwise the same state?
            const int pa = (lane ^ 16) << 2;
            const long long om = __double_as_longlong(cpm_bperm_f64(pa, m));
            const int orr = __builtin_amdgcn_ds_bpermute(pa, r);
            const uint64_t oh = cpm_bperm_u64(pa, hist);
            const bool diff = active && (om != __double_as_longlong(m) || orr != r || ((oh ^ hist) & hmask) != 0ull);
            return __builtin_amdgcn_ballot_w64(diff) == 0ull;
        };
        int done = 0;
        bool merged = false;
        for (int b = 0; b < nbatch && !merged; b += 2) {
            batch(b, pend0);
            done = (b + 1) * CPM_TB;
            merged = met();
            if (!merged) {
                batch(b + 1, pend1);
                done = (b + 2) * CPM_TB;
                merged = met();
            }
        }
        if (g == 1)                                                     // the new trajectory's decisions up to the meeting point
            for (int q = s; q < done; q += 16)
                if (k_first + q < P.ncalls) out[k_first + q] = dec[q];
        if (!merged) {                                                  // the chunk ENDS in another state than before
            if (g == 1 && active) {
                erec[48 + 3 * s] = (uint64_t)__double_as_longlong(m);
                erec[48 + 3 * s + 1] = (uint64_t)(int64_t)r;
                erec[48 + 3 * s + 2] = hist;
                if (state && k_first + P.CH >= P.ncalls) {              // ... and it owns the burst's last call: the carry
                    state[CPM_ST_STAGE + CPM_ST_M + s] = (uint64_t)__double_as_longlong(m);
                    state[CPM_ST_STAGE + CPM_ST_V + s] = (uint64_t)(int64_t)r;
                    state[CPM_ST_STAGE + CPM_ST_H + s] = hist;
                }
            }
            if (lane == 0 && chunk + 1 < P.nchunks)                     // the next chunk's start no longer matches: next round
                next_list[atomicAdd(reinterpret_cast<unsigned long long *>(next_count), 1ull)] = (uint64_t)(chunk + 1);
        }
        if (lane == 0) {
            atomicAdd(unmerged + 1, 1ull);                              // [1]: chunk repairs run, [2]: ... that handed on
            if (!merged) atomicAdd(unmerged + 2, 1ull);
        }
        return;
    }
    for (int b = 0; b < nbatch; b += 2) {
        batch(b, pend0);
        batch(b + 1, pend1);
    }
    // flush decisions: 16 B per lane and 256 calls
    if (live) {
        for (int off = 16 * s; off < P.CH; off += 256) {
            const int64_t k = k_first + off;
            if (k + 16 <= P.ncalls) {
                *reinterpret_cast<uint4 *>(out + k) = *reinterpret_cast<const uint4 *>(dec + off);
            } else {
                for (int q = 0; q < 16 && k + q < P.ncalls; ++q) out[k + q] = dec[off + q];
            }
        }
    }
    // proof record: what this chunk ended with
    if (live) {
        erec[48 + 3 * s] = (uint64_t)__double_as_longlong(m);
        erec[48 + 3 * s + 1] = (uint64_t)(int64_t)r;
        erec[48 + 3 * s + 2] = hist;
    }
    if (state && live && k_first + P.CH >= P.ncalls) {                 // the group that owns the last call
        state[CPM_ST_STAGE + CPM_ST_N] = (uint64_t)(n0 + P.ncalls);
        state[CPM_ST_STAGE + CPM_ST_M + s] = (uint64_t)__double_as_longlong(m);
        state[CPM_ST_STAGE + CPM_ST_V + s] = (uint64_t)(int64_t)r;
        state[CPM_ST_STAGE + CPM_ST_H + s] = hist;
    }
}

// rotation table as two 8 B columns (cos | sin): 2p <= 128 distinct entries land in distinct banks (equal entries are
// broadcast); as 16 B pairs entries 16 apart shared their banks.  The caller synchronises.
__device__ __forceinline__ double *cpm_stage_rot(const double2 *__restrict__ rot_cs, const cpm_vit_params &P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *rot = reinterpret_cast<double *>(smem + P.rot_off);
    for (int k = threadIdx.x; k < 2 * P.p; k += CPM_THREADS) {
        const double2 e = rot_cs[k];
        rot[k] = e.x;
        rot[CPM_ROT_SIN + k] = e.y;
    }
    return rot;
}

template <int M_, int LP_>
__global__ __launch_bounds__(CPM_THREADS, CPM_MIN_WAVES(M_, LP_)) void cpm_viterbi_kernel(const double2 *__restrict__ rows,
                                                                  const double2 *__restrict__ rot_cs,
                                                                  uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                                  uint64_t *__restrict__ edge,
                                                                  unsigned long long *__restrict__ unmerged,
                                                                  cpm_vit_params P)
{
    const double *rot = cpm_stage_rot(rot_cs, P);
    if (blockIdx.x == 0 && threadIdx.x < CPM_NLIST) cpm_list_counts(edge, P.nchunks, CPM_EDGE_WORDS)[threadIdx.x] = 0;   // the repair lists: empty
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = ((int64_t)blockIdx.x * CPM_WAVES + wave) * CPM_GROUPS + (lane >> 4);
    cpm_viterbi_body<M_, LP_, false>(rows, rot, out, state, edge, unmerged, P, chunk, nullptr, nullptr);
}

// One repair round (see wf_cpm_detect.h): the chunks of list `lin`, a wave each, grid-stride; chunks handed on go to list
// `lout`.  finisher != 0: ONE workgroup that keeps going, round after round between lists `lin` and `lout`, until a
// round hands nothing on — each round's smallest chunk is repaired from the true state, so the rounds end (at worst
// they are the sequential detector).  The waves of the finisher see each other's records and list entries through the
// agent-scope fences around the workgroup barrier; parallel rounds are separated by kernel boundaries.
template <int M_, int LP_>
__global__ __launch_bounds__(CPM_THREADS) void cpm_repair_kernel(const double2 *__restrict__ rows,
                                                                 const double2 *__restrict__ rot_cs,
                                                                 uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                                 uint64_t *__restrict__ edge,
                                                                 unsigned long long *__restrict__ unmerged,
                                                                 cpm_vit_params P, int lin, int lout, int finisher)
{
    uint64_t *const counts = cpm_list_counts(edge, P.nchunks, CPM_EDGE_WORDS);
    int64_t n = (int64_t)__hip_atomic_load(&counts[lin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == 0) return;                                               // (the whole grid: nothing listed)
    const double *rot = cpm_stage_rot(rot_cs, P);
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int64_t first = finisher ? wave : (int64_t)blockIdx.x * CPM_WAVES + wave, stride = finisher ? CPM_WAVES : (int64_t)gridDim.x * CPM_WAVES;
    for (;;) {
        const uint64_t *list = cpm_list(edge, P.nchunks, CPM_EDGE_WORDS, lin);
        for (int64_t idx = first; idx < n; idx += stride)
            cpm_viterbi_body<M_, LP_, true>(rows, rot, out, state, edge, unmerged, P,
                                            (int64_t)__hip_atomic_load(&list[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), &counts[lout],
                                            cpm_list(edge, P.nchunks, CPM_EDGE_WORDS, lout));
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

// Every chunk against its predecessor: thread = (chunk c >= 1, state s); a chunk that did NOT start from bitwise the
// (metric, phase index, last D decisions) chunk c - 1 ended with is LISTED for the repair launches (list 0), or — repair
// = 0: WF_OPT_DET_REPAIR off, and the closing check of WF_OPT_DET_FINAL_VERIFY — counted as unproven.
__global__ void cpm_verify_kernel(uint64_t *__restrict__ edge, int64_t nchunks, int S, uint64_t hmask,
                                  unsigned long long *__restrict__ unmerged, int repair)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c = idx / 16 + 1;
    const int s = (int)(idx & 15);
    bool bad = false;
    if (c < nchunks && s < S) {
        const uint64_t *a = edge + c * CPM_EDGE_WORDS, *b = edge + (c - 1) * CPM_EDGE_WORDS + 48;
        bad = a[3 * s] != b[3 * s] || a[3 * s + 1] != b[3 * s + 1] || ((a[3 * s + 2] ^ b[3 * s + 2]) & hmask) != 0ull;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(bad);
    const int lane = threadIdx.x & 63;
    if (s == 0 && ((m >> (lane & 48)) & 0xFFFFull)) {
        if (repair) {
            unsigned long long *counts = reinterpret_cast<unsigned long long *>(cpm_list_counts(edge, nchunks, CPM_EDGE_WORDS));
            cpm_list(edge, nchunks, CPM_EDGE_WORDS, 0)[atomicAdd(counts, 1ull)] = (uint64_t)c;
        } else {
            atomicAdd(unmerged, 1ull);
        }
    }
}

__global__ void cpm_carry_commit_kernel(uint64_t *state)
{
    const int t = threadIdx.x;
    if (t < 64) state[t] = state[CPM_ST_STAGE + t];
}

// Host: the trellis permutation.  Branches are enumerated exactly like cpm_oracle.c (the sequential statement kept with the tests)
// (start state ascending, then input ascending), so slot j of an end state is the j-th listed
// branch into it and strict '<' over slots 0, 1, ... reproduces that statement's tie-break.
static int cpm_build_tables(const wf_cpm_detector_config *d, cpm_vit_params &P)
{
    const int M = d->M, Lp = d->Lp, NC = d->NC, p = d->p;
    WF_REQUIRE((M == 2 || M == 4) && Lp >= 1 && Lp <= 3 && (d->nh == 1 || d->nh == 2) && p >= 1 && p <= 64 && NC >= 1 &&
                   p % NC == 0 && d->D >= 1,
               "wf_cpm: unsupported detector (M %d Lp %d nh %d p %d NC %d D %d)", M, Lp, d->nh, p, NC, d->D);
    const int lgM = M == 4 ? 2 : 1;
    WF_REQUIRE(d->D * lgM <= 64, "wf_cpm: decision delay %d does not fit the 64-bit decision register", d->D);
    int ncorr = 1, msub = 1, NF = 1;
    for (int i = 1; i < Lp; ++i) ncorr *= M;
    for (int i = 2; i < Lp; ++i) msub *= M;
    for (int i = 0; i < Lp; ++i) NF *= M;
    const int S = NC * ncorr;
    WF_REQUIRE(S <= 16, "wf_cpm: %d states (one DPP row holds 16, the wide form 64, the quad form 256 for pulses of 2 or 3 symbols)", S);
    for (int i = 0; i < d->nh; ++i) WF_REQUIRE(d->K[i] >= 0 && d->K[i] < p, "wf_cpm: K[%d] = %d outside [0, p)", i, d->K[i]);
    P.M = M; P.lgM = lgM; P.p = p; P.nh = d->nh; P.K0 = d->K[0]; P.K1 = d->nh == 2 ? d->K[1] : d->K[0];
    P.Lp = Lp; P.NC = NC; P.D = d->D; P.S = S; P.NF = NF;
    memset(&P.T, 0xFF, sizeof P.T);
    for (int kv = 0; kv < 3; ++kv) {
        const int K_old = kv == 2 ? 0 : (kv == 1 ? P.K1 : P.K0);
        int fill[16] = {0};
        for (int s = 0; s < S; ++s) {
            const int cls = s % NC, corr = s / NC;
            for (int u = 0; u < M; ++u) {
                const int u_old = Lp == 1 ? u : corr / msub;
                const int corr2 = Lp == 1 ? 0 : u + M * (corr % msub);
                const int inc = (K_old * u_old) % p;
                const int s2 = (cls + inc) % NC + NC * corr2;      // (v + inc) mod p mod NC == (cls + inc) mod NC: NC | p
                const int j = fill[s2]++;
                WF_REQUIRE(j < M, "wf_cpm: internal: more than M branches into a state");
                P.T.dest[kv][s][u] = (uint8_t)(4 * s2 + j);
                const int delta = ((2 * inc - (M - 1) * K_old) % (2 * p) + 2 * p) % (2 * p);
                P.T.info[kv][s2][j] = (uint16_t)(s | (u << 4) | (delta << 8));
            }
        }
        for (int s = 0; s < S; ++s) WF_REQUIRE(fill[s] == M, "wf_cpm: internal: state %d has %d incoming branches", s, fill[s]);
    }
    return WF_OK;
}

// Calls per chunk, and which form runs them.  Row form: a multiple of 64, at least 256 (and 2 W) so that the warm-up stays
// a fraction of the work, otherwise the smallest that puts the whole burst into ONE round of resident workgroups.  Lane
// form (wf_cpm_lanes.hip, where a specialisation for this trellis is compiled in): 64 chunks per wave, the chunk length that
// puts the burst into one round of the waves a CU's LDS holds, not below the specialisation's shortest worthwhile chunk
// (cpm_lane_plan::min_chunk).  A chunk that misses its warm-up and does not meet the first trajectory inside its own calls
// hands its repair on to the next chunk (wf_cpm_detect.h): chunk length and warm-up are matters of speed only.
// Which one: a lane runs its chunk alone, so the lane form's time is (chunk + warm-up) x its time per call whatever the
// burst's length, while the row form's falls with the burst (4 chunks per wave: 16 times the waves).  The two meet near
// 9e6 (ARTM) and 6.5e6 (PCM/FM) calls; below, the row form runs (a 2^22-call stream chunk: 0.32 against 0.54 ms for
// ARTM).  WF_OPT_CPM_FORM = 1 / 2 forces one form (tests run both on every size), WF_OPT_CPM_CHUNK_CALLS the chunk length.
// mf_form 1 | 2 (the lanes run the matched filters, alone | beside a front end): the lane form whatever the burst, and its own
// chunk rule.  That kernel is bound by vector issue, one wave saturating its SIMD (profiles/r06_mf_chunk_sweep.log) — alone, the
// chunk that puts ONE wave on every SIMD in a single round (1e7 calls: 160, 977 waves; 0.58 ms against 0.66 at 192, 0.80 at 256);
// beside the next block's front end both kernels share every SIMD's issue slots, the sum of their instructions is what counts and
// a third longer chunk does 5 % less warm-up work (pipelined link, same box: 160 0.930 | 192 0.908 | 208 0.834 | 224 0.843 | 256 0.873 ms).
static int64_t cpm_chunk_calls(const wf_ctx *ctx, const wf_cpm_detector_config *det, int64_t ncalls, int W, int wg_per_cu, cpm_lane_plan *lanes,
                               bool *use_lanes, int mf_form = 0)
{
    auto floor_ch = [&](int64_t ch, int min_chunk) {
        ch = (ch + 63) / 64 * 64;
        if (ch < min_chunk) ch = min_chunk;                        // (a matter of speed: the warm-up's share of the work, and most repairs ending inside their chunk)
        if (ch < 2 * W) ch = (2 * W + 63) / 64 * 64;
        return ch;
    };
    const int cus = ctx->cus;
    const int64_t form = ctx->opt[WF_OPT_CPM_FORM];
    const int64_t slots_row = (int64_t)cus * wg_per_cu * CPM_WAVES * CPM_GROUPS;
    int64_t ch = floor_ch((ncalls + slots_row - 1) / slots_row, 256);
    *use_lanes = false;
    // (a burst of fewer calls than the pulse has symbols is all virtual pre-start symbols: the lane form runs those
    // outside its call loop and has no chunk to hang the end record on — the row form takes it)
    if (form != 1 && ncalls >= det->Lp && wf_cpm_lanes_plan(det, lanes) == 0) {
        const int64_t slots_lane = (int64_t)cus * lanes->waves_per_cu * 64;
        const int64_t ch_lane = floor_ch((ncalls + slots_lane - 1) / slots_lane, lanes->min_chunk);
        const double t_lane = lanes->lane_ns_per_call * (double)(ch_lane + W), t_row = lanes->row_ns_per_call * (double)ncalls;
        if (form == 2 || mf_form || t_lane < t_row) {
            *use_lanes = true;
            ch = ch_lane;
        }
        if (mf_form) {
            const int64_t slots = (int64_t)cus * 4 * 64;             // one wave per SIMD
            ch = ((ncalls + slots - 1) / slots + 15) / 16 * 16;
            if (mf_form == 2) ch = (ch * 13 / 10 + 15) / 16 * 16;
            if (ch < 64) ch = 64;
            if (ch < 2 * W) ch = (2 * W + 15) / 16 * 16;
        }
    }
    if (ctx->opt[WF_OPT_CPM_CHUNK_CALLS] > 0) {                    // tuning aid (tools/cpm_vit_time.py)
        ch = (ctx->opt[WF_OPT_CPM_CHUNK_CALLS] + 15) / 16 * 16;   // (any multiple of 16, at least 64: the row form batches 8 calls, a lane stores 8 decisions at a time)
        if (ch < 64) ch = 64;
        if (ch <= W + 1) ch = (W + 2 + 63) / 64 * 64;
    }
    if (ch > 8192) ch = 8192;                                      // decision strips live in LDS; longer bursts take several rounds
    return ch;
}

static int cpm_warmup_calls(int warmup)                            // rows/lanes forms: 0 = the default, a multiple of two batches, <= 4096
{
    int W = warmup ? warmup : 96;
    W = (W + 2 * CPM_TB - 1) / (2 * CPM_TB) * (2 * CPM_TB);
    return W > 4096 ? 4096 : W;
}

// Which form of the detector wf_cpm_viterbi_detect runs for this trellis on this context (bench.py and the profile tools
// name the kernel they price by it): info4 = {form (0: row form, one 16-lane DPP row per chunk; 1: lane form, one lane per
// chunk; 2: wide form, one wave per chunk; 3: quad form, one workgroup per chunk), ring slots of the lane form, calls per chunk, warm-up calls} — computed by the
// functions the launch itself uses.
extern "C" int wf_cpm_detector_form(wf_ctx *ctx, const wf_cpm_detector_config *det, int64_t ncalls, int warmup, int *info4)
{
    WF_REQUIRE(ctx && det && info4 && ncalls >= 0 && warmup >= 0, "wf_cpm_detector_form: bad argument");
    if (wf_cpm_quad_applies(det)) {                                // 65 .. 256 states: thread = state, one workgroup per chunk
        const int W = wf_cpm_quad_warmup(warmup);
        info4[0] = 3;
        info4[1] = 0;
        info4[2] = (int)wf_cpm_quad_chunk_calls(ncalls, W, ctx->cus, ctx->opt[WF_OPT_CPM_CHUNK_CALLS]);
        info4[3] = W;
        return WF_OK;
    }
    if (wf_cpm_wide_applies(det)) {                                // 17 .. 64 states: lane = state, one wave per chunk
        const int W = wf_cpm_wide_warmup(warmup);
        info4[0] = 2;
        info4[1] = 0;
        info4[2] = (int)wf_cpm_wide_chunk_calls(ncalls, W, ctx->cus, ctx->opt[WF_OPT_CPM_CHUNK_CALLS]);
        info4[3] = W;
        return WF_OK;
    }
    cpm_vit_params P;
    const int rc = cpm_build_tables(det, P);
    if (rc) return rc;
    const int W = cpm_warmup_calls(warmup);
    cpm_lane_plan lanes{};
    bool use_lanes = false;
    const int64_t ch = cpm_chunk_calls(ctx, det, ncalls, W, CPM_MIN_WAVES(P.M, P.Lp), &lanes, &use_lanes);
    info4[0] = use_lanes ? 1 : 0;
    info4[1] = use_lanes ? lanes.ring_batches : 0;
    info4[2] = (int)ch;
    info4[3] = W;
    return WF_OK;
}

extern "C" int wf_cpm_viterbi_detect(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs,
                                     const double *d_rows_ri, int64_t ncalls, int warmup, uint8_t *d_decisions,
                                     void *d_state, void *stream)
{
    return wf_cpm_viterbi_detect_in(ctx, det, d_rot_cs, d_rows_ri, ncalls, warmup, d_decisions, d_state, stream, 0, 0);
}

// Would a launch with the matched filters inside the detector (cpm_mf_source) serve this burst?  The 16-filter ARTM design in
// its lane form, 9-tap templates at 8 samples per symbol whose windows start inside the sample array, and a burst the lane form
// would be chosen for anyway (a short burst runs the row form on rows: wf_cpm_viterbi_detect_in decides the same way).
bool wf_cpm_samples_form_applies(wf_ctx *ctx, const wf_cpm_detector_config *det, int64_t ncalls, int warmup, int sps, int nfilt, int ntm, int64_t start0)
{
    if (!ctx || !det || sps != 8 || nfilt != 16 || ntm != 9 || start0 < 0 || ncalls < det->Lp) return false;
    if (wf_cpm_quad_applies(det) || wf_cpm_wide_applies(det) || ctx->opt[WF_OPT_CPM_FORM] == 1) return false;
    cpm_lane_plan lanes{};
    if (wf_cpm_lanes_plan(det, &lanes) != 0 || lanes.spec != 0) return false;
    // one lane per chunk: on short bursts the chunks that fill the chip are mostly warm-up, and the row form on rows is the faster path
    // (1e9-symbol stream, same box: chunks of 2^22 calls 7.09 Gsym/s in this form against 7.35 on rows, chunks of 10 x 2^20 9.66 against 8.21)
    const int64_t floor_calls = ctx->opt[WF_OPT_CPM_SAMPLES_MIN_CALLS] ? ctx->opt[WF_OPT_CPM_SAMPLES_MIN_CALLS] : 6000000;
    return ncalls >= floor_calls;
}

// calls per chunk of such a launch (what wf_cpm_viterbi_detect_in will choose)
int wf_cpm_samples_form_chunk(wf_ctx *ctx, const wf_cpm_detector_config *det, int64_t ncalls, int warmup, bool beside)
{
    cpm_lane_plan lanes{};
    bool use_lanes = false;
    return (int)cpm_chunk_calls(ctx, det, ncalls, cpm_warmup_calls(warmup), 4, &lanes, &use_lanes, beside ? 2 : 1);
}

int wf_cpm_viterbi_detect_in(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri, int64_t ncalls,
                             int warmup, uint8_t *d_decisions, void *d_state, void *stream, int64_t slack_lo_bytes, int64_t slack_hi_bytes, bool beside,
                             const cpm_mf_source *mf, int edge_slot)
{
    WF_REQUIRE(ctx && det && ncalls >= 0 && warmup >= 0, "wf_cpm_viterbi_detect: bad argument");
    WF_REQUIRE(!mf || wf_cpm_quad_applies(det) || (!wf_cpm_wide_applies(det) && ncalls >= det->Lp && ctx->opt[WF_OPT_CPM_FORM] != 1),
               "wf_cpm_viterbi_detect: the matched-filter form is the lane form of a trellis of <= 16 states, or the quad form");
    const bool quad = wf_cpm_quad_applies(det) != 0;               // 65 .. 256 states: wf_cpm_quad.hip
    const bool wide = quad || wf_cpm_wide_applies(det) != 0;       // 17 .. 64 states: wf_cpm_wide.hip
    cpm_vit_params P;
    if (wide) {
        WF_REQUIRE((det->nh == 1 || det->nh == 2) && det->p >= 1 && det->p <= 64 && det->p % det->NC == 0 && det->D >= 1 &&
                       det->D * (det->M == 4 ? 2 : 1) <= 64,
                   "wf_cpm: unsupported detector (M %d Lp %d nh %d p %d NC %d D %d)", det->M, det->Lp, det->nh, det->p, det->NC, det->D);
        for (int i = 0; i < det->nh; ++i) WF_REQUIRE(det->K[i] >= 0 && det->K[i] < det->p, "wf_cpm: K[%d] = %d outside [0, p)", i, det->K[i]);
    }
    int rc = wide ? WF_OK : cpm_build_tables(det, P);
    if (rc) return rc;
    if (ncalls == 0) return WF_OK;
    WF_REQUIRE(d_rot_cs && d_rows_ri && d_decisions, "wf_cpm_viterbi_detect: NULL device pointer");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_rows_ri) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_decisions) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_rot_cs) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_state) & 15) == 0,
               "wf_cpm_viterbi_detect: device pointers must be 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    if (quad) return wf_cpm_quad_detect(ctx, det, d_rot_cs, d_rows_ri, ncalls, warmup, d_decisions, d_state, stream, mf);
    if (wide) return wf_cpm_wide_detect(ctx, det, d_rot_cs, d_rows_ri, ncalls, warmup, d_decisions, d_state, stream);
    // Default warm-up.  A chunk that misses its warm-up is REPAIRED by the launches behind the first (cpm_repair_kernel),
    // so the default is sized for the typical merge depth of the trellis, not for its tail, and backed by a scan at
    // 0 .. 12 dB (tools/cpm_warmup_scan.py, profiles/r03_cpm_repair_scan_*.json: 1.25e6 chunks per point): with
    // 96 calls ARTM's 16 states leave 1.3 % / 0.4 % / 0.04 % of the chunks to the repair at 0 / 4 / 6 dB (none from
    // 10 dB up) and binary PCM/FM 2.6 % / 0.1 % / 0.05 % at 0 / 4 / 10 dB; detector time 0.69 - 0.75 ms (ARTM) and
    // 0.49 - 0.54 ms (PCM/FM) per 1e7 calls.  A caller that knows its operating point may pass less
    // (waveforms_amd.link.operating_point_warmup: ARTM 48 from 8 dB up, PCM/FM 64): the warm-up sets how many chunks go
    // to the repair, never what the decisions are.
    const int W = cpm_warmup_calls(warmup);
    // Calls per chunk (a multiple of 64): at least 256 (and 2 W), so the warm-up stays a fraction of
    // the work, and otherwise the smallest that puts the whole burst into ONE round of resident
    // workgroups (4 per CU at this kernel's LDS / register use): with 512 calls per chunk 1e7 calls
    // made 1221 workgroups for 1024 slots — a second, almost empty round of the full chain length.
    // The lane-per-chunk form (wf_cpm_lanes.hip) where a specialisation for this trellis is compiled in: 64 chunks per
    // wave, as many waves as its LDS ring lets a CU hold — the chunk length that puts the burst into one round of them.
    cpm_lane_plan lanes{};
    const int wg_per_cu = CPM_MIN_WAVES(P.M, P.Lp);                 // resident workgroups per CU = waves per SIMD (4 waves per workgroup)
    bool use_lanes = false;
    const int64_t ch = cpm_chunk_calls(ctx, det, ncalls, W, wg_per_cu, &lanes, &use_lanes, mf ? (beside ? 2 : 1) : 0);
    WF_REQUIRE(!mf || (use_lanes && lanes.spec == 0), "wf_cpm_viterbi_detect: no lane specialisation runs the matched filters of this trellis");
    P.mf_templ = mf ? mf->d_templates : nullptr;
    P.mf_nsamp = mf ? mf->nsamp : 0;
    P.mf_col0 = mf ? (mf->col0 & 1) : 0;
    // (measured at 1e7 ARTM calls, W = 128, row form: 384 calls per chunk 1.07 ms, 512: 1.03, 640: 1.00, 768: 1.19,
    //  1024: 1.28, 1536: 1.68 — longer chunks do less warm-up work but leave fewer waves to hide the
    //  dependent chain of a call)
    P.CH = (int)ch;
    P.W = W;
    P.ncalls = ncalls;
    const int pieces = CPM_TB * P.NF;
    P.rows_off = 0;
    P.xch_off = CPM_GROUPS * pieces * 16;
    P.dec_off = P.xch_off + 4 * CPM_XS * 8;
    P.wave_bytes = (P.dec_off + CPM_GROUPS * P.CH + 15) / 16 * 16;
    P.rot_off = CPM_WAVES * P.wave_bytes;
    const size_t lds = (size_t)P.rot_off + (size_t)2 * CPM_ROT_SIN * 8;
    WF_REQUIRE(lds <= 160 * 1024, "wf_cpm_viterbi_detect: chunk of %d calls does not fit LDS", P.CH);
    const int64_t nchunks = (ncalls + P.CH - 1) / P.CH;
    const int64_t nwaves = (nchunks + CPM_GROUPS - 1) / CPM_GROUPS;
    const int64_t nblocks = (nwaves + CPM_WAVES - 1) / CPM_WAVES;
    WF_REQUIRE(nblocks < (1ll << 31), "wf_cpm_viterbi_detect: burst too long for one launch");
    P.nchunks = nchunks;
    // (edge_slot 0 / 1: a pipelined link runs consecutive blocks' detectors on two side streams, each with its own proof records)
    const size_t edge_words = (cpm_edge_total_words(nchunks, CPM_EDGE_WORDS) + 31) / 32 * 32;
    rc = wf_ctx_reserve_vit(ctx, edge_slot >= 0 ? 2 * edge_words : edge_words);
    if (rc) return rc;
    uint64_t *edge = reinterpret_cast<uint64_t *>(ctx->d_vit_edge) + (edge_slot > 0 ? edge_words : 0);
    hipStream_t s = wf_stream(stream);
    using kern_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, unsigned long long *, cpm_vit_params);
    using repair_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, unsigned long long *, cpm_vit_params, int, int, int);
    if (use_lanes) {
        rc = wf_cpm_lanes_launch(lanes, det, d_rot_cs, d_rows_ri, ncalls, W, P.CH, nchunks, d_decisions, d_state, edge, stream, slack_lo_bytes, slack_hi_bytes, !beside, mf);
        if (rc) return rc;
    } else {
        kern_t k = nullptr;
        if (P.M == 4) k = P.Lp == 1 ? cpm_viterbi_kernel<4, 1> : (P.Lp == 2 ? cpm_viterbi_kernel<4, 2> : cpm_viterbi_kernel<4, 3>);
        else k = P.Lp == 1 ? cpm_viterbi_kernel<2, 1> : (P.Lp == 2 ? cpm_viterbi_kernel<2, 2> : cpm_viterbi_kernel<2, 3>);
        if (lds > 48 * 1024)
            WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(CPM_THREADS), lds, s, reinterpret_cast<const double2 *>(d_rows_ri),
                           reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), edge,
                           ctx->d_vit_unmerged, P);
        WF_LAUNCH_CHECK();
    }
    if (nchunks > 1) {
        const uint64_t hmask = P.lgM * P.D >= 64 ? ~0ull : ((1ull << (P.lgM * P.D)) - 1ull);   // only the D decisions still inside the register can reach an output
        const int repair = ctx->opt[WF_OPT_DET_REPAIR] == 0 ? 1 : 0;
        const dim3 vgrid((unsigned)(((nchunks - 1) * 16 + 255) / 256));
        hipLaunchKernelGGL(cpm_verify_kernel, vgrid, dim3(256), 0, s, edge, nchunks, P.S, hmask, ctx->d_vit_unmerged, repair);
        WF_LAUNCH_CHECK();
        if (repair) {
            // chunks whose proof failed: their own calls again from the true state until both trajectories meet; a chunk
            // whose END changed hands on to the next one — two parallel rounds, then one workgroup that goes on until a
            // round hands nothing on (every wave of these launches leaves at once when nothing is listed)
            repair_t kr = nullptr;
            if (P.M == 4) kr = P.Lp == 1 ? cpm_repair_kernel<4, 1> : (P.Lp == 2 ? cpm_repair_kernel<4, 2> : cpm_repair_kernel<4, 3>);
            else kr = P.Lp == 1 ? cpm_repair_kernel<2, 1> : (P.Lp == 2 ? cpm_repair_kernel<2, 2> : cpm_repair_kernel<2, 3>);
            if (lds > 48 * 1024)
                WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            cpm_vit_params Pr = P;
            Pr.W = 0;
            for (int round = 0; round < 3; ++round) {
                hipLaunchKernelGGL(kr, dim3(round < 2 ? CPM_REPAIR_BLOCKS : 1), dim3(CPM_THREADS), lds, s, reinterpret_cast<const double2 *>(d_rows_ri),
                                   reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), edge,
                                   ctx->d_vit_unmerged, Pr, round, round + 1, round == 2 ? 1 : 0);
                WF_LAUNCH_CHECK();
            }
            if (ctx->opt[WF_OPT_DET_FINAL_VERIFY]) {
                hipLaunchKernelGGL(cpm_verify_kernel, vgrid, dim3(256), 0, s, edge, nchunks, P.S, hmask, ctx->d_vit_unmerged, 0);
                WF_LAUNCH_CHECK();
            }
        }
    }
    if (d_state) {
        hipLaunchKernelGGL(cpm_carry_commit_kernel, dim3(1), dim3(64), 0, s, static_cast<uint64_t *>(d_state));
        WF_LAUNCH_CHECK();
    }
    return WF_OK;
}

static int cpm_check_paired_templates(wf_ctx *ctx, const double *d_templates, int nh, int nfilt, int ntm, void *stream);

// wf_cpm_mf_rows_c128 + wf_cpm_viterbi_detect in ONE launch: the detector's lanes run the matched filters on the noisy samples
// (128 B per call instead of a 256 B row written and read back).  Returns 1 — not an error, nothing launched — when this form does
// not serve the configuration (wf_cpm_samples_form_applies); the templates must pair off as conjugates, f <-> nfilt - 1 - f (checked).
extern "C" int wf_cpm_viterbi_detect_samples(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_templates_ri,
                                             int nh, int nfilt, int ntm, const double *d_samples_ri, int64_t nsamp, int64_t start0, int sps,
                                             int64_t ncalls, int warmup, uint8_t *d_decisions, void *d_state, void *stream)
{
    WF_REQUIRE(ctx && det && ncalls >= 0 && warmup >= 0 && nsamp >= 0, "wf_cpm_viterbi_detect_samples: bad argument");
    WF_REQUIRE(nh == det->nh, "wf_cpm_viterbi_detect_samples: %d template columns for %d modulation indices", nh, det->nh);
    if (!wf_cpm_samples_form_applies(ctx, det, ncalls, warmup, sps, nfilt, ntm, start0)) return 1;
    WF_REQUIRE(d_rot_cs && d_templates_ri && d_samples_ri && d_decisions, "wf_cpm_viterbi_detect_samples: NULL device pointer");
    WF_REQUIRE(start0 + 8 * (ncalls - 1) + 8 < nsamp, "wf_cpm_viterbi_detect_samples: %lld calls from sample %lld run past %lld samples",
               (long long)ncalls, (long long)start0, (long long)nsamp);
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_samples_ri) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_templates_ri) & 15) == 0,
               "wf_cpm_viterbi_detect_samples: device pointers must be 16-byte aligned");
    int rc = cpm_check_paired_templates(ctx, d_templates_ri, nh, nfilt, ntm, stream);
    if (rc) return rc;
    const cpm_mf_source mf{d_templates_ri, nsamp - start0, 0};
    return wf_cpm_viterbi_detect_in(ctx, det, d_rot_cs, d_samples_ri + 2 * start0, ncalls, warmup, d_decisions, d_state, stream, 0, 0, false, &mf);
}

// ------------------------------------------------------------------------------------------
// Matched-filter rows: row n, filter f = sum_k r[start0 + n sps + k] * conj(T[n % nh][f][k]).
// A workgroup stages the samples of CPM_MF_SYMS consecutive symbols in LDS (each sample is read
// from HBM once); thread t owns filter f = t % NF for the symbols sub, sub + 256/NF, ...; its
// templates for both columns stay in registers (NTM = 9: the sps = 8 case), or come from an LDS
// copy (any length).  Accumulation order is that of cpm_oracle.c (the sequential statement kept with the tests):orc_cpm_mf_rows.
// NOISE: the channel of wf_awgn_c128 (derotation + Philox AWGN, same arithmetic) is applied while
// staging, so the link needs no separate 2.56 GB read-modify-write pass over the samples; the
// kernel is HBM-bound on its rows (16 * NF B written per symbol) and has the vector pipe to spare.
#define CPM_MF_SYMS 256

struct cpm_mf_params {
    int64_t nsamp, start0, ncalls;
    int sps, ntm, nh;
    int syms;                       // symbols per block (<= CPM_MF_SYMS)
    double rot_re, rot_im, sigma;   // NOISE: staged sample = r * rot + sigma * N(first_index + index)
    uint64_t seed, stream_id, first_index;
};

template <int NF, int NTM, bool NOISE>
__global__ __launch_bounds__(256) void cpm_mf_rows_kernel(const double2 *__restrict__ r, const double2 *__restrict__ templ,
                                                            double2 *__restrict__ out, cpm_mf_params P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *s_r = reinterpret_cast<double2 *>(smem);
    const int ntm = NTM ? NTM : P.ntm;
    const int span = P.syms * P.sps + ntm;                            // samples staged per block (last ones overlap the next block)
    double2 *s_t = s_r + span + 2;                                    // generic path: nh x NF x ntm templates
    __shared__ double2 s_tab[NOISE ? 256 : 1];                        // Gaussian source tables (see wf_tabs_lds)
    if (NOISE) wf_stage_tables<1, 0>(s_tab, threadIdx.x, 256);        // first use is behind the loop's barrier
    const int t = threadIdx.x;
    const int f = t % NF, sub = t / NF;
    constexpr int SUBS = 256 / NF;
    double2 tap[2][NTM ? NTM : 1];
    if (NTM) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < NTM; ++k) tap[c][k] = templ[((c < P.nh ? c : 0) * NF + f) * NTM + k];
    } else {
        for (int k = t; k < P.nh * NF * ntm; k += 256) s_t[k] = templ[k];
    }
    const int64_t nblk = (P.ncalls + P.syms - 1) / P.syms;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t nb = blk * P.syms;
        const int64_t s0 = P.start0 + nb * P.sps;
        __syncthreads();
        if (NOISE) {
            // one Philox block per PAIR of absolute sample indices (2q, 2q + 1), as wf_awgn_c128 draws them
            const int64_t a0 = (int64_t)P.first_index + s0;
            const int odd = (int)(a0 & 1);
            const uint64_t pair_lo = (uint64_t)(a0 >> 1);
            for (int q = t; 2 * q - odd < span; q += 256) {
                const int i0 = 2 * q - odd, i1 = i0 + 1;
                const int64_t b0 = s0 + i0, b1 = b0 + 1;
                const bool in0 = i0 >= 0 && b0 >= 0 && b0 < P.nsamp, in1 = i1 < span && b1 >= 0 && b1 < P.nsamp;
                const double2 c0 = in0 ? wf_load16_nt(r + b0) : make_double2(0.0, 0.0);
                const double2 c1 = in1 ? wf_load16_nt(r + b1) : make_double2(0.0, 0.0);
                double g[4];
                wf_gaussian_two(pair_lo + (uint64_t)q, P.stream_id, wf_opaque_seed(P.seed), P.sigma, wf_tabs_lds<1, 0>{s_tab}, g);
                if (i0 >= 0)
                    s_r[i0] = in0 ? make_double2(fma(c0.x, P.rot_re, fma(-c0.y, P.rot_im, g[0])), fma(c0.x, P.rot_im, fma(c0.y, P.rot_re, g[1])))
                                  : make_double2(0.0, 0.0);
                if (i1 < span)
                    s_r[i1] = in1 ? make_double2(fma(c1.x, P.rot_re, fma(-c1.y, P.rot_im, g[2])), fma(c1.x, P.rot_im, fma(c1.y, P.rot_re, g[3])))
                                  : make_double2(0.0, 0.0);
            }
        } else {
            for (int i = t; i < span; i += 256) {
                const int64_t a = s0 + i;
                s_r[i] = (a >= 0 && a < P.nsamp) ? wf_load16_nt(r + a) : make_double2(0.0, 0.0);
            }
        }
        __syncthreads();
        for (int sym = sub; sym < P.syms; sym += SUBS) {
            const int64_t n = nb + sym;
            if (n >= P.ncalls) break;
            const int c = (int)(n % P.nh);
            const double2 *x = s_r + sym * P.sps;
            double zr = 0.0, zi = 0.0;
            if (NTM) {
#pragma unroll
                for (int k = 0; k < NTM; ++k) {
                    const double2 xv = x[k];
                    const double2 tp = c ? tap[1][k] : tap[0][k];
                    zr = fma(xv.x, tp.x, fma(xv.y, tp.y, zr));
                    zi = fma(-xv.x, tp.y, fma(xv.y, tp.x, zi));      // (imaginary sample's term first in both sums: cpm_oracle.c)
                }
            } else {
                const double2 *tp = s_t + (c * NF + f) * ntm;
                for (int k = 0; k < ntm; ++k) {
                    const double2 xv = x[k];
                    zr = fma(xv.x, tp[k].x, fma(xv.y, tp[k].y, zr));
                    zi = fma(-xv.x, tp[k].y, fma(xv.y, tp[k].x, zi));
                }
            }
            out[n * NF + f] = make_double2(zr, zi);
        }
    }
}

static int cpm_mf_rows_launch(wf_ctx *ctx, const double *d_r_ri, int64_t nsamp, const double *d_templates_ri, int nh,
                              int nfilt, int ntm, int64_t start0, int sps, int64_t ncalls, double *d_rows_ri,
                              void *stream, bool noise, double rot_re, double rot_im, double sigma, uint64_t seed,
                              uint64_t stream_id, uint64_t first_index)
{
    WF_REQUIRE(ctx && nsamp >= 0 && ncalls >= 0, "wf_cpm_mf_rows_c128: bad argument");
    WF_REQUIRE((nh == 1 || nh == 2) && ntm >= 1 && ntm <= 257 && sps >= 1 && sps <= 256,
               "wf_cpm_mf_rows_c128: nh %d ntm %d sps %d", nh, ntm, sps);
    WF_REQUIRE(nfilt == 2 || nfilt == 4 || nfilt == 8 || nfilt == 16 || nfilt == 64,
               "wf_cpm_mf_rows_c128: %d filters per symbol (M^Lp with M in {2, 4}, Lp <= 3)", nfilt);
    if (ncalls == 0) return WF_OK;
    WF_REQUIRE(d_r_ri && d_templates_ri && d_rows_ri, "wf_cpm_mf_rows_c128: NULL device pointer");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_r_ri) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_rows_ri) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_templates_ri) & 15) == 0,
               "wf_cpm_mf_rows_c128: device pointers must be 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    cpm_mf_params P{nsamp, start0, ncalls, sps, ntm, nh, CPM_MF_SYMS, rot_re, rot_im, sigma, seed, stream_id, first_index};
    // the channel is generated in trips of 512 samples (one pair per thread): give up a few symbols
    // per block when that saves a ragged extra trip (2057 samples = 5 trips for one pair)
    if (noise)
        while (P.syms > 16 && (P.syms * sps + ntm + 1 + 511) / 512 > (P.syms * sps + 511) / 512 && (P.syms * sps + ntm + 1) % 512 < 64) --P.syms;
    const bool fast = ntm == 9;
    const size_t lds = ((size_t)P.syms * sps + ntm + 2 + (fast ? 0 : (size_t)nh * nfilt * ntm)) * sizeof(double2);
    WF_REQUIRE(lds <= 156 * 1024, "wf_cpm_mf_rows_c128: sps %d / %d-tap filters do not fit LDS staging", sps, ntm);
    const int64_t nblk = (ncalls + P.syms - 1) / P.syms;
    const int grid = (int)(nblk < 8192 ? nblk : 8192);
    using kern_t = void (*)(const double2 *, const double2 *, double2 *, cpm_mf_params);
    kern_t k;
#define CPM_MF_PICK(NFV) (noise ? (fast ? cpm_mf_rows_kernel<NFV, 9, true> : cpm_mf_rows_kernel<NFV, 0, true>) \
                               : (fast ? cpm_mf_rows_kernel<NFV, 9, false> : cpm_mf_rows_kernel<NFV, 0, false>))
    switch (nfilt) {
    case 2: k = CPM_MF_PICK(2); break;
    case 4: k = CPM_MF_PICK(4); break;
    case 8: k = CPM_MF_PICK(8); break;
    case 16: k = CPM_MF_PICK(16); break;
    default: k = CPM_MF_PICK(64); break;
    }
#undef CPM_MF_PICK
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, wf_stream(stream), reinterpret_cast<const double2 *>(d_r_ri),
                       reinterpret_cast<const double2 *>(d_templates_ri), reinterpret_cast<double2 *>(d_rows_ri), P);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

extern "C" int wf_cpm_mf_rows_c128(wf_ctx *ctx, const double *d_r_ri, int64_t nsamp, const double *d_templates_ri, int nh,
                                   int nfilt, int ntm, int64_t start0, int sps, int64_t ncalls, double *d_rows_ri,
                                   void *stream)
{
    return cpm_mf_rows_launch(ctx, d_r_ri, nsamp, d_templates_ri, nh, nfilt, ntm, start0, sps, ncalls, d_rows_ri, stream, false,
                              1.0, 0.0, 0.0, 0, 0, 0);
}

extern "C" int wf_cpm_awgn_mf_rows_c128(wf_ctx *ctx, const double *d_signal_ri, int64_t nsamp, double rot_re, double rot_im,
                                        double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                                        const double *d_templates_ri, int nh, int nfilt, int ntm, int64_t start0, int sps,
                                        int64_t ncalls, double *d_rows_ri, void *stream)
{
    return cpm_mf_rows_launch(ctx, d_signal_ri, nsamp, d_templates_ri, nh, nfilt, ntm, start0, sps, ncalls, d_rows_ri, stream, true,
                              rot_re, rot_im, sigma, seed, stream_id, first_index);
}

// ------------------------------------------------------------------------------------------
// 16 symbols per thread and trip (the arrays may start at any byte offset: gfx950 global loads need no
// alignment), one pair of atomics per workgroup — the first form loaded a byte per thread and issued
// 4096 same-address atomics: 54 us for 1e7 symbols against 14 us for the SOQPSK counter.
__global__ __launch_bounds__(256) void cpm_count_kernel(const uint8_t *__restrict__ dec, const int8_t *__restrict__ alpha, int M,
                                                        int64_t m, unsigned long long *__restrict__ counts)
{
    __shared__ long long s_part[2][4];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    long long se = 0, be = 0;
    const unsigned long long addm = 0x0101010101010101ull * (unsigned)(M - 1);
    const int64_t nvec = m / 16;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        ulonglong2 d, a;
        __builtin_memcpy(&d, dec + 16 * v, 16);
        __builtin_memcpy(&a, alpha + 16 * v, 16);
        // u = (alpha + M - 1) >> 1 per byte.  alpha is a signed byte, so the add is done on the low 7 bits with
        // the sign bit folded back in (no carry leaves a byte: 0x7F + 3 < 0x100); the mask after the shift drops
        // the bit that came in from the byte above
        auto x_of = [&](unsigned long long dd, unsigned long long aa) {
            const unsigned long long sum = ((aa & 0x7F7F7F7F7F7F7F7Full) + addm) ^ (aa & 0x8080808080808080ull);
            return dd ^ ((sum >> 1) & 0x0303030303030303ull);
        };
        const unsigned long long x0 = x_of(d.x, a.x), x1 = x_of(d.y, a.y);
        be += __popcll(x0) + __popcll(x1);
        auto nz = [](unsigned long long x) {       // symbols are 2 bits wide
            x |= x >> 1;
            return __popcll(x & 0x0101010101010101ull);
        };
        se += nz(x0) + nz(x1);
    }
    for (int64_t k = 16 * nvec + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
        const int u = ((int)alpha[k] + (M - 1)) >> 1;
        const int x = (int)dec[k] ^ u;
        se += x != 0;
        be += __popc(x);
    }
    se = wf_wave_sum_i64(se);
    be = wf_wave_sum_i64(be);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_part[0][wave] = se;
        s_part[1][wave] = be;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long a = 0, b = 0;
        for (int w = 0; w < 4; ++w) {
            a += s_part[0][w];
            b += s_part[1][w];
        }
        if (a) atomicAdd(&counts[0], (unsigned long long)a);
        if (b) atomicAdd(&counts[1], (unsigned long long)b);
    }
}

extern "C" int wf_cpm_count_errors(wf_ctx *ctx, const uint8_t *d_decided_u, const int8_t *d_ref_alpha, int M, int64_t m,
                                   int64_t *d_counts, void *stream)
{
    WF_REQUIRE(ctx && (M == 2 || M == 4) && m >= 0 && d_counts, "wf_cpm_count_errors: bad argument");
    if (m == 0) return WF_OK;
    WF_REQUIRE(d_decided_u && d_ref_alpha, "wf_cpm_count_errors: NULL device pointer");
    WF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(cpm_count_kernel, dim3(wf_grid_for(m, 256 * 16 * 8, 256)) /* <= 512 same-address atomics */, dim3(256), 0, wf_stream(stream), d_decided_u,
                       d_ref_alpha, M, m, reinterpret_cast<unsigned long long *>(d_counts));
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// fuse bit 6 promises that template f and template nfilt - 1 - f are exact conjugates (a symmetric alphabet makes them so):
// the front end then forms each pair from four real sums.  Checked here, value for value, on a host copy (wf_promise_verified).
struct cpm_pair_shape {
    int nh, nfilt, ntm;
};
static bool cpm_templates_pair_off(const unsigned char *const *host, const size_t *, const void *arg)
{
    const cpm_pair_shape *sh = static_cast<const cpm_pair_shape *>(arg);
    const double *t = reinterpret_cast<const double *>(host[0]);
    for (int c = 0; c < sh->nh; ++c)
        for (int f = 0; f < sh->nfilt; ++f)
            for (int k = 0; k < sh->ntm; ++k) {
                const double *a = t + 2 * (((size_t)c * sh->nfilt + f) * sh->ntm + k);
                const double *b = t + 2 * (((size_t)c * sh->nfilt + (sh->nfilt - 1 - f)) * sh->ntm + k);
                if (!(a[0] == b[0] && a[1] == -b[1])) return false;
            }
    return true;
}
static int cpm_check_paired_templates(wf_ctx *ctx, const double *d_templates, int nh, int nfilt, int ntm, void *stream)
{
    const cpm_pair_shape sh{nh, nfilt, ntm};
    const void *ptrs[1] = {d_templates};
    const size_t nb[1] = {(size_t)sh.nh * nfilt * ntm * 16};
    return wf_promise_verified(ctx, 1, ptrs, nb, 1, stream, cpm_templates_pair_off, &sh,
                               "templates f and nfilt - 1 - f are not exact conjugates of each other (wf_cpm_link_config.fuse bit 6 / the samples form of the detector)");
}
static int cpm_check_paired(wf_ctx *ctx, const wf_cpm_link_config *cfg, int nfilt, int ntm, void *stream)
{
    return cpm_check_paired_templates(ctx, cfg->d_templates, cfg->det.nh, nfilt, ntm, stream);
}

// ------------------------------------------------------------------------------------------
// Device-resident link for multi-h CPM / PCM/FM: the SOQPSK link's structure (wf_pipeline.hip)
// with the mapper, the multi-index modulator and the generic detector.
struct cpm_link_layout {
    int64_t nsym, nbits, npts, ncalls, start0;
    int ntm, nfilt, bps;
    size_t off_bits, off_syms, off_sig, off_rows, off_dec, total;
};

static inline int64_t cpm_round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static bool cpm_make_layout(const wf_cpm_link_config *cfg, cpm_link_layout &L)
{
    if (!cfg || cfg->nsym < 1 || cfg->sps < 2 || cfg->sps > 256 || cfg->ntaps < 1) return false;
    if (cfg->mapper_kind != 1 && cfg->mapper_kind != 2) return false;
    const wf_cpm_detector_config &d = cfg->det;
    if ((d.M != 2 && d.M != 4) || d.Lp < 1 || d.Lp > 3) return false;
    L.bps = cfg->mapper_kind == 1 ? 2 : 1;
    if ((L.bps == 2) != (d.M == 4)) return false;
    L.nsym = cfg->nsym;
    L.nbits = cfg->nsym * L.bps;
    L.npts = wf_fir_out_len(cfg->nsym, cfg->sps, cfg->ntaps);
    L.ntm = cfg->sps + 1;
    // cpm_detect.py:cpm_geometry of the test-side statement
    const int c = (cfg->ntaps - 1) / 2;
    const int span = (cfg->ntaps - 1) - d.Lp * cfg->sps;
    const int o = span >= 0 ? span / 2 : -((-span + 1) / 2);          // floor division
    if (o < -(cfg->sps / 2)) return false;
    L.start0 = cfg->sps - c + o;
    int64_t nc = (L.npts - L.ntm - L.start0) / cfg->sps + 1;
    if (L.npts - L.ntm - L.start0 < 0) nc = 0;
    L.ncalls = nc < cfg->nsym ? nc : cfg->nsym;
    L.nfilt = 1;
    for (int i = 0; i < d.Lp; ++i) L.nfilt *= d.M;
    size_t o_ = 0;
    L.off_bits = o_; o_ += (size_t)cpm_round_up(L.nbits + 16, 256);
    L.off_syms = o_; o_ += (size_t)cpm_round_up(L.nsym + 16, 256);
    L.off_sig = o_;  o_ += (size_t)cpm_round_up(L.npts * 16, 256);
    L.off_rows = o_; o_ += (size_t)cpm_round_up(L.ncalls * L.nfilt * 16, 256);
    L.off_dec = o_;  o_ += (size_t)cpm_round_up(L.ncalls + 16, 256);
    L.total = o_;
    return true;
}

extern "C" int64_t wf_cpm_link_workspace_bytes(const wf_cpm_link_config *cfg)
{
    cpm_link_layout L;
    if (!cpm_make_layout(cfg, L)) return -1;
    const int64_t one = ((int64_t)L.total + 255) / 256 * 256;
    return (cfg->fuse & 32) ? 2 * one : (int64_t)L.total;      // fuse bit 5: two sets of intermediates (see wf_cpm_link_run)
}

extern "C" int wf_cpm_link_layout(const wf_cpm_link_config *cfg, int64_t *info8)
{
    cpm_link_layout L;
    if (!info8 || !cpm_make_layout(cfg, L)) return WF_ERR_VALUE;
    info8[0] = L.ncalls; info8[1] = L.start0; info8[2] = (int64_t)L.off_dec; info8[3] = (int64_t)L.off_syms;
    info8[4] = (int64_t)L.off_sig; info8[6] = L.npts; info8[7] = (int64_t)L.off_rows;
    // [5]: 1 when fuse bits 1 + 3 will run modulator + channel + filters as ONE kernel for this configuration
    info8[5] = ((cfg->fuse & 8) && (cfg->fuse & 2) && L.ncalls > 0 &&
                wf_mod_chan_cpm_rows_applies(cfg->nsym, cfg->det.nh, cfg->ntaps, cfg->sps, L.nfilt, L.ntm, L.start0)) ? 1 : 0;
    return WF_OK;
}

// fuse bit 7 (with bits 1 and 6): the front end stores the noisy SAMPLES and the detector's lanes run the matched filters —
// where both kernels serve the configuration (16 paired templates, 8 samples per symbol, a burst the lane form takes)
// ... and the 64-filter, 256-state design (quad form: the workgroup's own threads form the filter outputs of a batch, the plain
// k-ascending chain — no promise about the templates needed, any start0)
static bool cpm_link_samples_form_quad(const wf_cpm_link_config *cfg, const cpm_link_layout &L)
{
    return (cfg->fuse & 128) && (cfg->fuse & 2) && L.ncalls > 0 && wf_cpm_quad_applies(&cfg->det) && L.nfilt == 64 && L.ntm == 9 && cfg->sps == 8 &&
           wf_mod_chan_samples_applies(cfg->nsym, cfg->det.nh, cfg->ntaps, cfg->sps);
}
static bool cpm_link_samples_form(wf_ctx *ctx, const wf_cpm_link_config *cfg, const cpm_link_layout &L)
{
    if (cpm_link_samples_form_quad(cfg, L)) return true;
    return (cfg->fuse & 128) && (cfg->fuse & 64) && (cfg->fuse & 2) && L.ncalls > 0 &&
           wf_mod_chan_samples_applies(cfg->nsym, cfg->det.nh, cfg->ntaps, cfg->sps) &&
           wf_cpm_samples_form_applies(ctx, &cfg->det, L.ncalls, cfg->warmup, cfg->sps, L.nfilt, L.ntm, L.start0);
}

// What wf_cpm_link_run will launch for this configuration on this context: info4[0] = front end (0: modulator, channel and
// matched filters as separate kernels; 1: one kernel, rows out; 2: modulator + channel in one kernel, SAMPLES out, the matched
// filters inside the detector), info4[1 .. 3] = detector form, calls per chunk, warm-up calls (wf_cpm_detector_form).
extern "C" int wf_cpm_link_form(wf_ctx *ctx, const wf_cpm_link_config *cfg, int *info4)
{
    cpm_link_layout L;
    WF_REQUIRE(ctx && info4 && cpm_make_layout(cfg, L), "wf_cpm_link_form: bad configuration");
    int d4[4] = {0, 0, 0, 0};
    const int rc = wf_cpm_detector_form(ctx, &cfg->det, L.ncalls, cfg->warmup, d4);
    if (rc) return rc;
    int64_t i8[8];
    wf_cpm_link_layout(cfg, i8);
    info4[0] = cpm_link_samples_form(ctx, cfg, L) ? 2 : (int)i8[5];
    info4[1] = d4[0]; info4[2] = d4[2]; info4[3] = d4[3];
    if (info4[0] == 2 && !cpm_link_samples_form_quad(cfg, L)) {
        info4[1] = 1;
        info4[2] = wf_cpm_samples_form_chunk(ctx, &cfg->det, L.ncalls, cfg->warmup, (cfg->fuse & 32) != 0);
    }
    return WF_OK;
}

extern "C" int wf_cpm_link_run(wf_ctx *ctx, const wf_cpm_link_config *cfg, void *d_workspace, int64_t workspace_bytes,
                               int64_t *d_counts, int64_t *h_compared, void *stream)
{
    WF_REQUIRE(ctx && cfg && d_workspace && d_counts, "wf_cpm_link_run: NULL argument");
    cpm_link_layout L;
    WF_REQUIRE(cpm_make_layout(cfg, L), "wf_cpm_link_run: bad configuration");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_workspace) & 255) == 0, "wf_cpm_link_run: workspace must be 256-byte aligned");
    WF_REQUIRE((int64_t)L.total <= workspace_bytes, "wf_cpm_link_run: workspace too small (%lld < %lld)",
               (long long)workspace_bytes, (long long)L.total);
    WF_REQUIRE(cfg->d_h && cfg->d_pulse && cfg->d_templates && cfg->d_rot_cs, "wf_cpm_link_run: NULL table pointer");
    char *w = static_cast<char *>(d_workspace);
    // fuse bit 5: the detector and the error count of a block on the context's side stream, beside the front end of the
    // next wf_cpm_link_run on this context (as wf_link_run does it: two sets of intermediates, used alternately)
    // (the 256-state samples form runs its blocks one after the other: its front end is 3 % of a block, and whatever shares the chip with
    //  the quad detector — 8 workgroups per CU, LDS-bound — slows it by more than it hides: 11.67 ms per block against 13.49 pipelined,
    //  profiles/r06_bench_multih256_forms.log)
    const bool piped = (cfg->fuse & 32) != 0 && L.ncalls > 0 && !cpm_link_samples_form_quad(cfg, L);
    if (piped) {
        const int64_t set_bytes = ((int64_t)L.total + 255) / 256 * 256;
        WF_REQUIRE(2 * set_bytes <= workspace_bytes, "wf_cpm_link_run: fuse bit 5 needs two sets of intermediates (%lld bytes)", (long long)(2 * set_bytes));
        WF_HIP(hipSetDevice(ctx->device));
#ifndef WF_CPM_PIPE_PRIO
#define WF_CPM_PIPE_PRIO 0      // (A/B aid: priority of the side streams, 0 = default; -1 = high: profiles/r06_ab_side_stream_priority.log)
#endif
        if (!ctx->pipe_stream) {
            hipStream_t ps;
            WF_HIP(hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, WF_CPM_PIPE_PRIO));
            ctx->pipe_stream = ps;
            WF_HIP(hipEventCreateWithFlags(&ctx->pipe_front, hipEventDisableTiming));
            for (int k = 0; k < 2; ++k) WF_HIP(hipEventCreateWithFlags(&ctx->pipe_done[k], hipEventDisableTiming));
        }
        if (!ctx->pipe_stream2) {
            hipStream_t ps;
            WF_HIP(hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, WF_CPM_PIPE_PRIO));
            ctx->pipe_stream2 = ps;
        }
        w += (int64_t)ctx->pipe_set * set_bytes;
        if (ctx->pipe_done_valid[ctx->pipe_set]) WF_HIP(hipStreamWaitEvent(wf_stream(stream), ctx->pipe_done[ctx->pipe_set], 0));
    } else {
        const int rj = wf_link_join_internal(ctx, stream);
        if (rj) return rj;
    }
    uint8_t *bits = reinterpret_cast<uint8_t *>(w + L.off_bits);
    int8_t *syms = reinterpret_cast<int8_t *>(w + L.off_syms);
    double *sig = reinterpret_cast<double *>(w + L.off_sig);
    double *rows = reinterpret_cast<double *>(w + L.off_rows);
    uint8_t *dec = reinterpret_cast<uint8_t *>(w + L.off_dec);
    hipEvent_t *ev = nullptr;
    if (cfg->event_slot >= 0) {
        WF_REQUIRE(cfg->event_slot < WF_LINK_EVENT_SLOTS, "wf_cpm_link_run: event_slot %d", cfg->event_slot);
        if (!ctx->events) {
            ctx->events = new hipEvent_t[WF_LINK_EVENT_SLOTS * (WF_LINK_STAGES + 1)];
            for (int k = 0; k < WF_LINK_EVENT_SLOTS * (WF_LINK_STAGES + 1); ++k) WF_HIP(hipEventCreate(&ctx->events[k]));
        }
        ev = ctx->events + cfg->event_slot * (WF_LINK_STAGES + 1);
    }
#define MARK(k) do { if (ev) WF_HIP(hipEventRecord(ev[k], wf_stream(stream))); } while (0)
    int rc;
    MARK(0);
    // PRBS and mapper in one launch (the mappers are memoryless: the PRBS kernel applies them to the bits it holds in LDS);
    // fuse bit 4 (16), as in wf_link_config: the two generic kernels instead (same bits and symbols)
    rc = (cfg->fuse & 16) ? 1 : wf_lfsr_generate_map(ctx, cfg->degree, cfg->mask, cfg->state, cfg->skip, bits, L.nbits, cfg->mapper_kind, syms, stream);
    if (rc < 0) return rc;
    if (rc == 1) {
        if ((rc = wf_lfsr_generate(ctx, cfg->degree, cfg->mask, cfg->state, cfg->skip, bits, L.nbits, nullptr, stream))) return rc;
        MARK(1);
        // MultiHSymbolMapper on an even number of bits leaves its parity at 0 (precoder.py:22)
        if ((rc = wf_symbol_map(ctx, cfg->mapper_kind, bits, L.nbits, 0, 0, 0, syms, stream))) return rc;
    } else {
        MARK(1);
    }
    MARK(2);
    // fuse bit 3 (with bit 1): modulator + channel + matched-filter rows in one kernel
    // (mod_chan_bank_kernel, wf_modulate.hip) — the baseband samples never reach HBM
    bool fused_all = false, samples_form = false;
    const bool samples_quad = cpm_link_samples_form_quad(cfg, L);
    if (cpm_link_samples_form(ctx, cfg, L)) {
        if (!samples_quad && (rc = cpm_check_paired(ctx, cfg, L.nfilt, L.ntm, stream))) return rc;
        rc = wf_mod_chan_samples(ctx, syms, cfg->nsym, cfg->d_h, cfg->det.nh, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, cos(-M_PI / 4), sin(-M_PI / 4),
                                 cfg->sigma, cfg->seed, cfg->stream_id, 0, sig, stream);
        if (rc < 0) return rc;
        samples_form = fused_all = rc == 0;
    }
    if (!fused_all && (cfg->fuse & 8) && (cfg->fuse & 2) && L.ncalls > 0) {
        wf_mcb_opts mo;
        mo.runs_hint = piped ? (L.nfilt == 4 ? 16 : 8) : 0;              // (the detector of the previous block shares the chip: finer runs)
        mo.cpm_paired = (cfg->fuse & 64) != 0;                           // fuse bit 6: templates f and nfilt - 1 - f are conjugates — checked
        if (mo.cpm_paired && (rc = cpm_check_paired(ctx, cfg, L.nfilt, L.ntm, stream))) return rc;
        rc = wf_mod_chan_cpm_rows(ctx, syms, cfg->nsym, cfg->d_h, cfg->det.nh, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, cfg->d_templates,
                                  L.nfilt, L.ntm, L.start0, cos(-M_PI / 4), sin(-M_PI / 4), cfg->sigma, cfg->seed, cfg->stream_id, L.ncalls,
                                  rows, stream, &mo);
        if (rc < 0) return rc;
        fused_all = rc == 0;
    }
    if (!fused_all) {
        rc = wf_cpm_modulate_c128(ctx, syms, cfg->nsym, cfg->d_h, cfg->det.nh, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, sig, stream);
        if (rc < 0) return rc;
        WF_REQUIRE(rc == 0, "wf_cpm_link_run: burst / pulse outside the fused modulator (%lld symbols, %d taps)",
                   (long long)cfg->nsym, cfg->ntaps);
    }
    MARK(3);
    MARK(4);
    if (fused_all) {
        MARK(5);
    } else if (cfg->fuse & 2) {     // channel inside the matched-filter kernel: the noisy samples never exist in HBM
        MARK(5);
        if ((rc = wf_cpm_awgn_mf_rows_c128(ctx, sig, L.npts, cos(-M_PI / 4), sin(-M_PI / 4), cfg->sigma, cfg->seed, cfg->stream_id, 0,
                                           cfg->d_templates, cfg->det.nh, L.nfilt, L.ntm, L.start0, cfg->sps, L.ncalls, rows, stream))) return rc;
    } else {
        if ((rc = wf_awgn_c128(ctx, sig, L.npts, cos(-M_PI / 4), sin(-M_PI / 4), cfg->sigma, cfg->seed, cfg->stream_id, 0, sig, stream))) return rc;
        MARK(5);
        if ((rc = wf_cpm_mf_rows_c128(ctx, sig, L.npts, cfg->d_templates, cfg->det.nh, L.nfilt, L.ntm, L.start0, cfg->sps, L.ncalls,
                                      rows, stream))) return rc;
    }
    void *back = stream;
    // A block's back end is detector, proof, up to three repair launches and the count: a serial chain of ~0.1 ms behind the
    // detector that the NEXT block's detector has no part in — so consecutive blocks' back ends take turns on TWO side streams
    // (own proof records each: edge_slot), and a detector starts when its samples are in, not when its predecessor's repairs are done.
    // (trellises above 16 states keep one stream: their forms hold one set of proof records per context)
    const bool two_back = piped && !wf_cpm_quad_applies(&cfg->det) && !wf_cpm_wide_applies(&cfg->det);
    const int edge_slot = two_back ? ctx->pipe_set : -1;
    if (piped) {
        back = two_back && ctx->pipe_set ? ctx->pipe_stream2 : ctx->pipe_stream;
        WF_HIP(hipEventRecord(ctx->pipe_front, wf_stream(stream)));
        WF_HIP(hipStreamWaitEvent(wf_stream(back), ctx->pipe_front, 0));
    }
#define MARKB(k) do { if (ev) WF_HIP(hipEventRecord(ev[k], wf_stream(back))); } while (0)
    MARKB(6);
    // (rows sit inside the block's set of intermediates: what lies before them — symbols, the sample region — and behind
    //  them — the decisions, up to the end of the set — may be read by the lane form's row fetch, so no wave of it clamps)
    if (samples_form && samples_quad) {
        const cpm_mf_source mf{cfg->d_templates, L.npts, 0, L.start0};
        if ((rc = wf_cpm_viterbi_detect_in(ctx, &cfg->det, cfg->d_rot_cs, sig, L.ncalls, cfg->warmup, dec, nullptr, back, 0, 0, piped, &mf, -1))) return rc;
    } else if (samples_form) {
        // (the sample array sits inside the set too: bits and symbols before it, the — unused — row region and the decisions behind)
        const cpm_mf_source mf{cfg->d_templates, L.npts - L.start0, 0};
        if ((rc = wf_cpm_viterbi_detect_in(ctx, &cfg->det, cfg->d_rot_cs, sig + 2 * L.start0, L.ncalls, cfg->warmup, dec, nullptr, back,
                                           (int64_t)L.off_sig + 16 * L.start0, (int64_t)(L.total - L.off_sig) - L.npts * 16, piped, &mf, edge_slot)))
            return rc;
    } else if ((rc = wf_cpm_viterbi_detect_in(ctx, &cfg->det, cfg->d_rot_cs, rows, L.ncalls, cfg->warmup, dec, nullptr, back, (int64_t)L.off_rows,
                                              (int64_t)(L.total - L.off_rows) - L.ncalls * L.nfilt * 16, piped, nullptr, edge_slot)))
        return rc;
    MARKB(7);
    // decision of call k is symbol k - D + 1; symbols [skip_head, ncalls - D] are compared
    const int64_t skip = cfg->skip_head > 0 ? cfg->skip_head : 0;
    int64_t m = L.ncalls - cfg->det.D + 1 - skip;
    if (m < 0) m = 0;
    if (m > 0)
        if ((rc = wf_cpm_count_errors(ctx, dec + skip + cfg->det.D - 1, syms + skip, cfg->det.M, m, d_counts, back))) return rc;
    MARKB(8);
#undef MARKB
    if (piped) {
        WF_HIP(hipEventRecord(ctx->pipe_done[ctx->pipe_set], wf_stream(back)));
        ctx->pipe_done_valid[ctx->pipe_set] = true;
        ctx->pipe_set ^= 1;
    }
#undef MARK
    if (h_compared) *h_compared = m;
    return WF_OK;
}

// ------------------------------------------------------------------------------------------
// Streaming form of the CPM link (BASELINE config 5's scheme, wf_pipeline.hip, for the waveforms of
// configs[2]): the same chain over a stream of cfg->nsym symbols in chunks of `chunk_symbols` detector
// calls with the HBM footprint of ONE chunk.  Chunk c makes calls [c B, (c+1) B); what it needs from its
// neighbours is (a) re-generated as a halo — PRBS by leap-ahead, mapper (memoryless), noise by absolute
// sample index, one modulator tile either side — or (b) carried in a device block of
// WF_CPM_STREAM_STATE_BYTES: the detector state (WF_CPM_STATE_BYTES: call counter, metrics, tilted phase
// indices, decision registers) and the 62-bit fixed-point phase carry of the next chunk's first tile.
// Decisions and counts equal the one-shot wf_cpm_link_run over the whole stream, bit for bit.  Runs the
// one-kernel front end (fuse bits 1 + 3; sps 8, 4 or 16 filters): other configurations are refused.
struct cpm_stream_layout {
    int64_t N, B, tile_len, spt, ntiles_total, halo, ncalls_total, m_total;
    int64_t k_lo, ncols, ws, nloc, tile_lo, ntiles, q_out_tile;
    size_t off_bits, off_syms, off_rows, off_dec, total;
    cpm_link_layout L;
    bool ok;
};

static cpm_stream_layout cpm_make_stream_layout(const wf_cpm_link_config *cfg, int64_t B, int64_t c)
{
    cpm_stream_layout S{};
    if (!cpm_make_layout(cfg, S.L) || B <= 0 || c < 0) return S;
    S.N = cfg->nsym; S.B = B;
    S.ok = wf_mod_tile_geometry(cfg->sps, cfg->ntaps, cfg->nsym, &S.tile_len, &S.spt, &S.ntiles_total) == 0;
    if (!S.ok || S.spt <= 0) { S.ok = false; return S; }
    S.halo = cpm_round_up(S.spt + 48 + cfg->det.D, 16);
    S.ncalls_total = S.L.ncalls;
    const int64_t skip = cfg->skip_head > 0 ? cfg->skip_head : 0;
    S.m_total = S.ncalls_total - cfg->det.D + 1 - skip;
    if (S.m_total < 0) S.m_total = 0;
    S.ok = B % S.spt == 0 && B % 128 == 0 && B >= 4 * S.halo &&
           ((cfg->fuse & 8) && (cfg->fuse & 2) &&
            wf_mod_chan_cpm_rows_applies(cfg->nsym, cfg->det.nh, cfg->ntaps, cfg->sps, S.L.nfilt, S.L.ntm, S.L.start0));
    S.k_lo = c * B;
    const int64_t k_hi = S.k_lo + B < S.ncalls_total ? S.k_lo + B : S.ncalls_total;
    S.ncols = k_hi > S.k_lo ? k_hi - S.k_lo : 0;
    S.ws = c * B - S.halo > 0 ? c * B - S.halo : 0;
    const int64_t we = (c + 1) * B + S.halo < S.N ? (c + 1) * B + S.halo : S.N;
    S.nloc = we > S.ws ? we - S.ws : 0;
    S.tile_lo = c * B / S.spt - 1 > 0 ? c * B / S.spt - 1 : 0;          // the tile before the chunk too: its last column is the chunk's first
    int64_t tile_hi = (c + 1) * B / S.spt + 1;
    if (tile_hi > S.ntiles_total) tile_hi = S.ntiles_total;
    S.ntiles = tile_hi > S.tile_lo ? tile_hi - S.tile_lo : 0;
    const int64_t next_tile_lo = (c + 1) * B / S.spt - 1;
    S.q_out_tile = (next_tile_lo >= S.tile_lo && next_tile_lo < tile_hi) ? next_tile_lo - S.tile_lo : -1;
    size_t o = 0;
    const int64_t win = B + 2 * S.halo + 32;
    S.off_bits = o; o += (size_t)cpm_round_up(win * S.L.bps, 256);
    S.off_syms = o; o += (size_t)cpm_round_up(win, 256);
    S.off_rows = o; o += (size_t)cpm_round_up(B * S.L.nfilt * 16, 256);
    S.off_dec = o;  o += (size_t)cpm_round_up(B + 16, 256);
    S.total = o;
    return S;
}

extern "C" int64_t wf_cpm_link_stream_workspace_bytes(const wf_cpm_link_config *cfg, int64_t chunk_symbols)
{
    const cpm_stream_layout S = cpm_make_stream_layout(cfg, chunk_symbols, 0);
    return S.ok ? (int64_t)S.total : -1;
}

/* info8 = {calls in the chunk, first call index, off(decisions), off(symbols alpha), global index of symbols[0],
 *          calls of the whole stream, symbols one modulator tile holds, off(rows)} */
extern "C" int wf_cpm_link_stream_layout(const wf_cpm_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index, int64_t *info8)
{
    if (!info8) return WF_ERR_VALUE;
    const cpm_stream_layout S = cpm_make_stream_layout(cfg, chunk_symbols, chunk_index);
    if (!S.ok) return WF_ERR_VALUE;
    info8[0] = S.ncols; info8[1] = S.k_lo; info8[2] = (int64_t)S.off_dec; info8[3] = (int64_t)S.off_syms;
    info8[4] = S.ws; info8[5] = S.ncalls_total; info8[6] = S.spt; info8[7] = (int64_t)S.off_rows;
    return WF_OK;
}

extern "C" int wf_cpm_link_stream_chunk(wf_ctx *ctx, const wf_cpm_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                                        void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                                        int64_t *h_compared, void *stream)
{
    return wf_cpm_link_stream_chunk_phase(ctx, cfg, chunk_symbols, chunk_index, d_state, d_workspace, workspace_bytes, d_counts, h_compared, 7, stream);
}

// Parts of a chunk for callers that pipeline chunks on two streams (own workspace AND own wf_ctx per stream), numbered
// like wf_link_stream_chunk_phase: bit 0 = PRBS + mapper (stateless: leap-ahead position, memoryless mapper — depends on
// nothing); bit 2 = modulator + channel + matched-filter rows (needs this chunk's bit-0 part and the phase carry the
// previous chunk's bit-2 part left in d_state); bit 1 = detector + error count (needs this chunk's bit-2 part and the
// detector carry of the previous chunk's bit-1 part).  phases = 7: the whole chunk.
extern "C" int wf_cpm_link_stream_chunk_phase(wf_ctx *ctx, const wf_cpm_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                                              void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                                              int64_t *h_compared, int phases, void *stream)
{
    WF_REQUIRE(phases >= 1 && phases <= 7, "wf_cpm_link_stream_chunk_phase: phases %d", phases);
    WF_REQUIRE(ctx && cfg && d_state && d_workspace && d_counts, "wf_cpm_link_stream_chunk: NULL argument");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_workspace) & 255) == 0 && (reinterpret_cast<uintptr_t>(d_state) & 15) == 0,
               "wf_cpm_link_stream_chunk: workspace must be 256-byte aligned, the carry block 16-byte aligned");
    const cpm_stream_layout S = cpm_make_stream_layout(cfg, chunk_symbols, chunk_index);
    WF_REQUIRE(S.ok, "wf_cpm_link_stream_chunk: chunk of %lld calls must be a multiple of the modulator tile (%lld symbols) and of 128 "
               "and at least 4 halos (%lld), and the configuration one the one-kernel front end takes (fuse bits 1 + 3, sps 8)",
               (long long)chunk_symbols, (long long)S.spt, (long long)(4 * S.halo));
    WF_REQUIRE((int64_t)S.total <= workspace_bytes, "wf_cpm_link_stream_chunk: workspace too small");
    WF_REQUIRE(cfg->d_h && cfg->d_pulse && cfg->d_templates && cfg->d_rot_cs, "wf_cpm_link_stream_chunk: NULL table pointer");
    if (h_compared) *h_compared = 0;
    if (S.ncols == 0) return WF_OK;
    char *w = static_cast<char *>(d_workspace);
    uint8_t *bits = reinterpret_cast<uint8_t *>(w + S.off_bits);
    int8_t *syms = reinterpret_cast<int8_t *>(w + S.off_syms);
    double *rows = reinterpret_cast<double *>(w + S.off_rows);
    uint8_t *dec = reinterpret_cast<uint8_t *>(w + S.off_dec);
    char *carry = static_cast<char *>(d_state);
    uint64_t *q_phase = reinterpret_cast<uint64_t *>(carry + WF_CPM_STATE_BYTES);
    const int bps = S.L.bps;
    int rc;
    if (phases & 1) {
        if ((rc = wf_lfsr_generate(ctx, cfg->degree, cfg->mask, cfg->state, cfg->skip + (uint64_t)(S.ws * bps), bits, S.nloc * bps, nullptr, stream)))
            return rc;
        // (the mappers are memoryless per symbol; a window starts on a symbol boundary, so the multi-h mapper's parity is 0)
        if ((rc = wf_symbol_map(ctx, cfg->mapper_kind, bits, S.nloc * bps, 0, 0, 0, syms, stream))) return rc;
    }
    // fuse bit 7 (as in wf_cpm_link_run): the chunk's noisy samples instead of its rows — they take the rows' place in the workspace
    // (128 B per call where a 16-filter row is 256) — and the matched filters inside the detector, where the chunk is one the lane
    // form takes (the last, shorter chunk of a stream may fall back to rows: both forms carry the same detector state, and the
    // modulator's tile window and phase carry are those of the rows form either way)
    const bool samples_form = (cfg->fuse & 128) && (cfg->fuse & 64) && (cfg->fuse & 2) && S.L.start0 == 0 &&
                              wf_mod_chan_samples_applies(cfg->nsym, cfg->det.nh, cfg->ntaps, cfg->sps) &&
                              wf_cpm_samples_form_applies(ctx, &cfg->det, S.ncols, cfg->warmup, cfg->sps, S.L.nfilt, S.L.ntm, S.L.start0);
    const int64_t smp_lo = 8 * S.k_lo;                                    // global index of the first sample the chunk's detector reads
    int64_t smp_hi = 8 * (S.k_lo + S.ncols) + 8;                          // ... one past the last (the last window's ninth sample, + padding to a whole call)
    if (smp_hi > S.L.npts) smp_hi = S.L.npts;
    if ((phases & 4) && samples_form) {
        if ((rc = cpm_check_paired(ctx, cfg, S.L.nfilt, S.L.ntm, stream))) return rc;
        rc = wf_mod_chan_samples_window(ctx, syms, S.ws, S.nloc, S.N, cfg->d_h, cfg->det.nh, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, S.tile_lo,
                                        S.ntiles, q_phase, q_phase, S.q_out_tile, cos(-M_PI / 4), sin(-M_PI / 4), cfg->sigma, cfg->seed,
                                        cfg->stream_id, 0, rows, smp_lo, smp_hi, stream);
        if (rc < 0) return rc;
        WF_REQUIRE(rc == 0, "wf_cpm_link_stream_chunk: internal: the samples front end refused the window");
    } else if (phases & 4) {
        wf_mcb_opts mo;
        mo.cpm_paired = (cfg->fuse & 64) != 0;
        if (mo.cpm_paired && (rc = cpm_check_paired(ctx, cfg, S.L.nfilt, S.L.ntm, stream))) return rc;
        rc = wf_mod_chan_bank_window(ctx, syms, S.ws, S.nloc, S.N, cfg->d_h, cfg->det.nh, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, S.tile_lo,
                                     S.ntiles, q_phase, q_phase, S.q_out_tile, cfg->d_templates, cos(-M_PI / 4), sin(-M_PI / 4), cfg->sigma,
                                     cfg->seed, cfg->stream_id, 0, nullptr, S.L.start0 + 4, S.k_lo, S.ncols, 0, rows, stream, S.L.nfilt,
                                     cfg->det.nh, 3, 0, &mo);
        if (rc < 0) return rc;
        WF_REQUIRE(rc == 0, "wf_cpm_link_stream_chunk: internal: the one-kernel front end refused the window");
    }
    if (phases & 2) {
        // (the rows sit inside the chunk's workspace: symbols before them, decisions behind — see wf_cpm_link_run)
        if (samples_form) {
            const cpm_mf_source mf{cfg->d_templates, smp_hi - smp_lo, (int)(S.k_lo & 1)};
            if ((rc = wf_cpm_viterbi_detect_in(ctx, &cfg->det, cfg->d_rot_cs, rows, S.ncols, cfg->warmup, dec, carry, stream, (int64_t)S.off_rows,
                                               (int64_t)(S.total - S.off_rows) - (smp_hi - smp_lo) * 16, phases != 7, &mf)))
                return rc;
        } else if ((rc = wf_cpm_viterbi_detect_in(ctx, &cfg->det, cfg->d_rot_cs, rows, S.ncols, cfg->warmup, dec, carry, stream, (int64_t)S.off_rows,
                                                  (int64_t)(S.total - S.off_rows) - S.ncols * S.L.nfilt * 16, phases != 7)))   // (issued in parts: chunks overlap on two streams)
            return rc;
        // decision of call k is symbol k - D + 1; symbols [skip_head, ncalls - D] of the stream are compared
        const int64_t skip = cfg->skip_head > 0 ? cfg->skip_head : 0;
        const int64_t k_first = skip + cfg->det.D - 1;                  // first call whose decision is compared
        const int64_t j0 = S.k_lo >= k_first ? 0 : k_first - S.k_lo;
        const int64_t ncmp = S.ncols - j0;
        if (ncmp > 0) {
            const int64_t sym0 = S.k_lo + j0 - cfg->det.D + 1;          // global index of the first reference symbol
            if ((rc = wf_cpm_count_errors(ctx, dec + j0, syms + (sym0 - S.ws), cfg->det.M, ncmp, d_counts, stream))) return rc;
            if (h_compared) *h_compared = ncmp;
        }
    }
    return WF_OK;
}
