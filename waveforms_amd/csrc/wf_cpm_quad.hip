// wf_cpm_quad.hip — the generic CPM trellis detector for trellises of 65 .. 256 states: thread = state, one WORKGROUP of
// four waves = one detector (wf_cpm_detect.hip: a 16-lane DPP row; wf_cpm_lanes.hip: a lane; wf_cpm_wide.hip: a wave).
//
// What it is for: the FULL trellis of ARTM multi-h CPM, N_S = p M^(L-1) = 16 * 4^2 = 256 states with the pulse kept to its
// three symbols (notes/cpm/cpm.md:128-140) and 64 matched filters per symbol — the yardstick the reduced 16- and 64-state
// designs are measured against (0.2 - 0.3 dB at BER 1e-3 .. 1e-5), on the GPU.  The algorithm is the one cpm_oracle.c
// defines (conventions of waveforms/viterbi/algorithm.py:57-98: increment Re(rotation * mf) minimised, strict '<' / first
// listed branch on ties, first arg-min, min-normalised metrics, one decision per call from the best state); decisions are
// bit-identical to it.
//
// Per call every state thread rotates its M matched-filter outputs by its survivor's phase and drops the M candidates into
// the LDS slots of the end states they lead to; after a workgroup barrier it reads its own M incoming candidates, picks the
// first minimum, and fetches the winner's phase index and decision register from LDS (the winner may live in another wave);
// the 256-state minimum is four wave minima (DPP + v_readlane) met in LDS behind a second barrier.  Two barriers per call.
// For a pulse of Lp >= 2 symbols the trellis permutation needs no table: the M branches into an end state come from the M
// start states that differ in the symbol LEAVING the window (u_old = 0 .. M-1), whose indices ascend with u_old — so slot j
// of an end state IS the branch with u_old = j, in the sequential statement's list order (start state ascending, then input).
// Chunk-parallel with the same proof and the same cascading repair as the other forms (wf_cpm_detect.h): a chunk starts
// `warmup` calls early, records the state its own calls started from and ended with (3 words per state), a small kernel
// compares neighbours bitwise, and the repair launches run listed chunks again as a PAIR of detectors in one workgroup of
// eight waves (one from the recorded start, one from the previous chunk's end) until they meet.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "wf_cpm_detect.h"

#define QUAD_TB 4               // calls per staged batch of rows
#define QUAD_T 256              // threads (states) per detector
#define QUAD_XS 260             // exchange: candidate j of end state e at word j * QUAD_XS + e (8 B words)
#define QUAD_EDGE_WORDS (2 * 256 * 3)
// Device-resident detector state, quad layout (inside WF_CPM_STATE_BYTES = 16384): [0] calls made, [1 + s] metrics,
// [257 + s] tilted phase indices, [513 + s] decision registers; staging copy from word 1024.
#define QUAD_ST_N 0
#define QUAD_ST_M 1
#define QUAD_ST_V 257
#define QUAD_ST_H 513
#define QUAD_ST_WORDS 769
#define QUAD_ST_STAGE 1024
static_assert((QUAD_ST_STAGE + QUAD_ST_WORDS) * 8 <= WF_CPM_STATE_BYTES, "the 256-state carry and its staging copy fit the state block");

struct cpm_quad_params {
    int M, p, nh, K0, K1, Lp, NC, D, S, NF, msub;
    int CH, W;
    int64_t ncalls, nchunks;
    int rows_off, xch_off, src_off, dec_off, min_off, cmp_off, team_bytes, rot_off, smp_off;   // dynamic LDS layout (bytes)
    // round 6: the matched filters inside the detector — `rows` are the noisy samples: a batch's 8 QUAD_TB + 1 samples are fetched
    // one per thread a batch ahead and staged in LDS, then thread s forms filter s % NF of call s / NF of the batch from the 9
    // samples of that call's window (the k-ascending chain of cpm_mf_rows_kernel and cpm_oracle.c, bit for bit): 36 multiply-adds
    // per thread and batch, one more workgroup barrier per batch, and 128 B per call from HBM where a row of 64 filter outputs is
    // 1 KB.  The taps of a thread are the same for every batch (batches start on even calls) and are re-read through L1 in
    // groups of three — held in registers they would halve the workgroups a CU holds.
    const double *mf_templ;
    int64_t mf_nsamp, mf_start0;
    int mf_col0;
};

__device__ __forceinline__ double quad_min_raw(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// min over the 64 lanes of a wave: row_ror 8, 4, 2, 1 inside each DPP row, then the four row minima through the scalar file
__device__ __forceinline__ double quad_wave_min(double v)
{
    v = quad_min_raw(v, wf_dpp_f64<0x128, 0xf>(v));
    v = quad_min_raw(v, wf_dpp_f64<0x124, 0xf>(v));
    v = quad_min_raw(v, wf_dpp_f64<0x122, 0xf>(v));
    v = quad_min_raw(v, wf_dpp_f64<0x121, 0xf>(v));
    const long long b = __double_as_longlong(v);
    const int lo = (int)b, hi = (int)(b >> 32);
    double q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = __builtin_amdgcn_readlane(lo, 16 * k), h = __builtin_amdgcn_readlane(hi, 16 * k);
        q[k] = __longlong_as_double(((long long)h << 32) | (unsigned)l);
    }
    return quad_min_raw(quad_min_raw(q[0], q[1]), quad_min_raw(q[2], q[3]));      // (no NaNs among metrics: min is exact and order-free; fmin would quiet each operand first)
}

// One detector = the QUAD_T threads of `team` (0, or 0 / 1 in a repair workgroup).  Every barrier below is a WORKGROUP
// barrier: in a repair both teams run the same calls and meet at the same barriers.
template <int M_, int LP_, bool REPAIR>
__device__ __forceinline__ void cpm_quad_body(const double2 *__restrict__ rows, const double *__restrict__ rot,
                                              uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                              uint64_t *__restrict__ edge, unsigned long long *__restrict__ unmerged,
                                              const cpm_quad_params &P, const int64_t chunk, uint64_t *__restrict__ next_count,
                                              uint64_t *__restrict__ next_list)
{
    constexpr int M = M_;
    constexpr int LGM = M_ == 4 ? 2 : 1;
    constexpr int NF = LP_ == 2 ? M_ * M_ : M_ * M_ * M_;
    constexpr int PIECES = QUAD_TB * NF;                      // 16 B pieces per batch
    constexpr int PL = (PIECES + QUAD_T - 1) / QUAD_T;        // pieces per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int team = REPAIR ? (int)(threadIdx.x >> 8) : 0;
    const int s = threadIdx.x & (QUAD_T - 1);
    const int wave = s >> 6, lane = s & 63;
    const bool active = s < P.S;
    const int cls = s % P.NC, corr = s / P.NC;
    char *tbase = smem + team * P.team_bytes;
    double2 *rowbuf = reinterpret_cast<double2 *>(tbase + P.rows_off);
    double *xch = reinterpret_cast<double *>(tbase + P.xch_off);          // [M slots][QUAD_XS]
    int *rsrc = reinterpret_cast<int *>(tbase + P.src_off);               // tilted phase index per state
    uint64_t *hsrc = reinterpret_cast<uint64_t *>(tbase + P.src_off + QUAD_T * 4);   // decision register per state
    const bool writes_out = !REPAIR || team == 1;                         // (a repair's team 0 re-runs what is already there)
    double *wmin = reinterpret_cast<double *>(tbase + P.min_off);         // the four waves' minima

    const int64_t n0 = state ? (int64_t)state[QUAD_ST_N] : 0;            // calls made before this launch
    const int64_t k_first = chunk * P.CH;                                 // first own call (local index)
    const bool live = k_first < P.ncalls;
    const int T = P.W + P.CH;

    // The trellis as arithmetic (Lp >= 2).  As a START state: its candidates land in slot u_old = the symbol leaving the window,
    // at end states base + NC u.  As an END state: slot j comes from the start state with u_old = j.
    const int msub = P.msub;
    const int u_old = corr / msub;
    const int u_new = corr % M;                                           // the newest symbol of every branch into this state
    auto tables = [&](int K_old, int &xbase, uint32_t &srcpk, uint32_t &delpk) __attribute__((always_inline)) {
        const int inc_out = (K_old * u_old) % P.p;
        xbase = u_old * QUAD_XS + (cls + inc_out) % P.NC + P.NC * M * (corr % msub);
        srcpk = 0;
        delpk = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const int inc = (K_old * j) % P.p;
            int sc = (cls - inc) % P.NC;
            sc += sc < 0 ? P.NC : 0;
            const int src = sc + P.NC * (j * msub + corr / M);
            const int delta = ((2 * inc - (M - 1) * K_old) % (2 * P.p) + 2 * P.p) % (2 * P.p);
            srcpk |= (uint32_t)src << (8 * j);
            delpk |= (uint32_t)delta << (8 * j);
        }
    };
    int xb[2];
    uint32_t srcp[2], delp[2];
    tables(P.K0, xb[0], srcp[0], delp[0]);
    tables(P.K1, xb[1], srcp[1], delp[1]);

    double m = active ? 0.0 : INFINITY;
    const int64_t k_start = chunk == 0 ? 0 : k_first - P.W;              // first call this detector really runs
    int r = 2 * cls - cpm_tilt(P.M, P.p, P.nh, P.K0, P.K1, P.Lp, n0 + k_start);
    r += r < 0 ? 2 * P.p : 0;
    uint64_t hist = 0;
    if (state && chunk == 0 && n0 > 0) {                                  // continue the carried detector
        m = active ? __longlong_as_double((long long)state[QUAD_ST_M + s]) : INFINITY;
        r = (int)state[QUAD_ST_V + s];
        hist = state[QUAD_ST_H + s];
    }
    uint64_t *const erec = edge + chunk * QUAD_EDGE_WORDS;
    if constexpr (REPAIR) {
        const uint64_t *src = team ? erec - QUAD_EDGE_WORDS / 2 : erec;   // team 1: the previous chunk's end (as it is now) | team 0: this chunk's recorded start
        const uint64_t w0 = active ? src[3 * s] : 0ull, w1 = active ? src[3 * s + 1] : 0ull, w2 = active ? src[3 * s + 2] : 0ull;
        m = active ? __longlong_as_double((long long)w0) : INFINITY;
        r = (int)w1;
        hist = w2;
        __syncthreads();                                                  // team 0 has the OLD start before team 1 replaces it
        if (team == 1 && active) {                                        // what this run starts from becomes the chunk's recorded start (wf_cpm_detect.h)
            erec[3 * s] = w0;
            erec[3 * s + 1] = w1;
            erec[3 * s + 2] = w2;
        }
    }

    auto fetch = [&](int b, double2 (&dst)[PL]) __attribute__((always_inline)) {
        const int64_t kb = k_first - P.W + (int64_t)b * QUAD_TB;          // local call of the batch's first row
        if (P.mf_templ) {
            // sample s of the batch's 8 QUAD_TB + 1 (threads beyond them fetch nothing); outside the burst: zero (wf_cpm_mf_rows_c128)
            const int64_t idx = P.mf_start0 + 8 * kb + s;
            const bool in = s <= 8 * QUAD_TB && idx >= 0 && idx < P.mf_nsamp;
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d v = {0.0, 0.0};
            if (in) v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(rows + idx));
            dst[0] = make_double2(v.x, v.y);
            return;
        }
#pragma unroll
        for (int i = 0; i < PL; ++i) {
            const int q = s + QUAD_T * i;
            const int qq = q < PIECES ? q : 0;
            int64_t row = kb + qq / NF;
            row = row < 0 ? 0 : (row >= P.ncalls ? P.ncalls - 1 : row);   // never decoded when clamped
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(rows + row * NF + qq % NF));
            dst[i] = make_double2(v.x, v.y);
        }
    };
    const double2 *zlane = rowbuf + M * corr;
    const int dshift = LGM * (P.D - 1);

    // One detector call; every thread of the workgroup goes through both barriers of every call.
    auto step = [&](int tt, int t, bool emit) __attribute__((always_inline)) {
        const int64_t k = k_first - P.W + t;                              // local call index
        const int64_t n = n0 + k;                                         // global call index
        const bool valid = live && k >= 0 && k < P.ncalls;                // (uniform over the team)
        const int64_t m_old = n - LP_ + 1;
        const int kv = m_old < 0 ? 2 : (P.nh == 2 ? (int)(m_old & 1) : 0);
        int xbase;
        uint32_t srcpk, delpk;
        if (kv == 2) tables(0, xbase, srcpk, delpk);                      // virtual pre-start symbols carry no phase (the first Lp - 1 calls of a burst)
        else {
            xbase = xb[kv];
            srcpk = srcp[kv];
            delpk = delp[kv];
        }
        if (active) {
            const double cr = rot[r], sr = rot[CPM_ROT_SIN + r];
            const double2 *zrow = zlane + tt * NF;
#pragma unroll
            for (int u = 0; u < M; ++u) {
                const double2 z = zrow[u];
                xch[xbase + P.NC * u] = m + (-fma(cr, z.x, sr * z.y));   // metric - Re(e^{-j theta} Z)
            }
            rsrc[s] = r;
            hsrc[s] = hist;
        }
        __syncthreads();
        double best = INFINITY;
        int jbest = 0;
        if (active) {
            double c[M];
#pragma unroll
            for (int j = 0; j < M; ++j) c[j] = xch[j * QUAD_XS + s];
            best = c[0];
#pragma unroll
            for (int j = 1; j < M; ++j) {                                 // strict '<' in list order: the first listed branch keeps a tie
                const bool lt = c[j] < best;
                best = lt ? c[j] : best;
                jbest = lt ? j : jbest;
            }
        }
        const int src = (int)((srcpk >> (8 * jbest)) & 0xFFu), delta = (int)((delpk >> (8 * jbest)) & 0xFFu);
        const uint32_t nr_raw = (uint32_t)(rsrc[active ? src : 0] + delta);                         // < 4p
        const int nr = (int)min(nr_raw, nr_raw - (uint32_t)(2 * P.p));                              // mod 2p
        const uint64_t nh_ = (hsrc[active ? src : 0] << LGM) | (uint64_t)u_new;
        const double wm = quad_wave_min(best);
        if (lane == 0) wmin[wave] = wm;
        __syncthreads();
        const double w0 = wmin[0], w1 = wmin[1], w2 = wmin[2], w3 = wmin[3];
        const double gmin = quad_min_raw(quad_min_raw(w0, w1), quad_min_raw(w2, w3));
        const double nm = best - gmin;                                    // the minimum becomes exactly 0.0
        if (valid) {
            m = nm;
            r = nr;
            hist = nh_;
        }
        if (emit && valid) {
            // np.argmin: the first state whose metric is the minimum = the first wave whose minimum is the global one,
            // and in it the first lane at 0.0
            const int wfirst = w0 == gmin ? 0 : (w1 == gmin ? 1 : (w2 == gmin ? 2 : 3));
            const unsigned long long zero = __builtin_amdgcn_ballot_w64(nm == 0.0);
            if (wave == wfirst && __builtin_amdgcn_inverse_ballot_w64(zero & (0ull - zero)))
                if (writes_out) out[k] = (n >= P.D - 1) ? (uint8_t)((nh_ >> dshift) & (uint64_t)(M - 1)) : (uint8_t)0;   // (one byte straight to the output: a strip in LDS cost two resident workgroups per CU)
        }
    };

    double2 pend[PL];
    const int nbatch = T / QUAD_TB;
    fetch(0, pend);
    double2 *smpbuf = reinterpret_cast<double2 *>(tbase + P.smp_off);     // MF form: the batch's 8 QUAD_TB + 1 samples
    auto batch = [&](int b) __attribute__((always_inline)) {
        if (P.mf_templ) {
            if (s <= 8 * QUAD_TB) smpbuf[s] = pend[0];
        } else {
#pragma unroll
            for (int i = 0; i < PL; ++i) {
                const int q = s + QUAD_T * i;
                if (q < PIECES) rowbuf[q] = pend[i];
            }
        }
        fetch(b + 1 < nbatch ? b + 1 : nbatch - 1, pend);                 // issued unconditionally
        const int t0 = b * QUAD_TB;
        if (!REPAIR && t0 == P.W && live && active) {                     // the next call is the chunk's first own one
            erec[3 * s] = (uint64_t)__double_as_longlong(m);
            erec[3 * s + 1] = (uint64_t)(int64_t)r;
            erec[3 * s + 2] = hist;
        }
        __syncthreads();                                                  // rows staged (and the previous batch consumed: its last call ended in a barrier)
        if (P.mf_templ) {
            // filter s % NF of call s / NF of the batch (NF = 64: one per thread; fewer filters: the first QUAD_TB NF threads)
            if (s < PIECES) {
                const int i = s / NF, f = s % NF;
                const int64_t row = k_first - P.W + (int64_t)b * QUAD_TB + i;
                const double2 *tp = reinterpret_cast<const double2 *>(P.mf_templ) + ((P.nh == 2 ? (int)((row + P.mf_col0) & 1) : 0) * NF + f) * 9;
                const double2 *xs = smpbuf + 8 * i;
                double zr = 0.0, zi = 0.0;
#pragma unroll 1
                for (int g = 0; g < 3; ++g) {
#pragma unroll
                    for (int k = 3 * g; k < 3 * g + 3; ++k) {
                        const double2 xv = xs[k], tk = tp[k];
                        zr = fma(xv.x, tk.x, fma(xv.y, tk.y, zr));
                        zi = fma(-xv.x, tk.y, fma(xv.y, tk.x, zi));              // (imaginary sample's term first in both sums: cpm_oracle.c)
                    }
                }
                rowbuf[s] = make_double2(zr, zi);
            }
            __syncthreads();
        }
        const bool emit = t0 >= P.W;
#pragma unroll 1
        for (int tt = 0; tt < QUAD_TB; ++tt) step(tt, t0 + tt, emit);
    };
    if constexpr (REPAIR) {
        // (P.W = 0 in this launch.)  The two teams compare their states through LDS after every batch.
        const uint64_t hmask = LGM * P.D >= 64 ? ~0ull : ((1ull << (LGM * P.D)) - 1ull);
        uint64_t *cmp = reinterpret_cast<uint64_t *>(smem + P.cmp_off);   // [2 teams][3][QUAD_T]
        int done = 0;
        bool merged = false;
        for (int b = 0; b < nbatch && !merged; ++b) {
            batch(b);
            done = (b + 1) * QUAD_TB;
            uint64_t *mine = cmp + team * 3 * QUAD_T, *theirs = cmp + (team ^ 1) * 3 * QUAD_T;
            mine[s] = (uint64_t)__double_as_longlong(m);
            mine[QUAD_T + s] = (uint64_t)(int64_t)r;
            mine[2 * QUAD_T + s] = hist & hmask;
            __syncthreads();
            const bool diff = active && (theirs[s] != (uint64_t)__double_as_longlong(m) || theirs[QUAD_T + s] != (uint64_t)(int64_t)r ||
                                         theirs[2 * QUAD_T + s] != (hist & hmask));
            merged = __syncthreads_or(diff ? 1 : 0) == 0;                 // (every thread of both teams gets the same answer)
        }
        (void)done;
        if (team == 1) {                                                  // (the new trajectory's decisions up to the meeting point are in place)
            if (!merged) {                                                // the chunk ENDS in another state than before
                if (active) {
                    erec[QUAD_EDGE_WORDS / 2 + 3 * s] = (uint64_t)__double_as_longlong(m);
                    erec[QUAD_EDGE_WORDS / 2 + 3 * s + 1] = (uint64_t)(int64_t)r;
                    erec[QUAD_EDGE_WORDS / 2 + 3 * s + 2] = hist;
                    if (state && k_first + P.CH >= P.ncalls) {            // ... and it owns the burst's last call: the carry
                        state[QUAD_ST_STAGE + QUAD_ST_M + s] = (uint64_t)__double_as_longlong(m);
                        state[QUAD_ST_STAGE + QUAD_ST_V + s] = (uint64_t)(int64_t)r;
                        state[QUAD_ST_STAGE + QUAD_ST_H + s] = hist;
                    }
                }
                if (s == 0 && chunk + 1 < P.nchunks)                      // the next chunk's start no longer matches: next round
                    next_list[atomicAdd(reinterpret_cast<unsigned long long *>(next_count), 1ull)] = (uint64_t)(chunk + 1);
            }
            if (s == 0) {
                atomicAdd(unmerged + 1, 1ull);                            // [1]: chunk repairs run, [2]: ... that handed on
                if (!merged) atomicAdd(unmerged + 2, 1ull);
            }
        }
        return;
    }
    for (int b = 0; b < nbatch; ++b) batch(b);
    if (live) {
        if (active) {                                                     // proof record: what this chunk ended with
            erec[QUAD_EDGE_WORDS / 2 + 3 * s] = (uint64_t)__double_as_longlong(m);
            erec[QUAD_EDGE_WORDS / 2 + 3 * s + 1] = (uint64_t)(int64_t)r;
            erec[QUAD_EDGE_WORDS / 2 + 3 * s + 2] = hist;
        }
    }
    if (state && live && k_first + P.CH >= P.ncalls) {                    // the detector that owns the last call
        if (s == 0) state[QUAD_ST_STAGE + QUAD_ST_N] = (uint64_t)(n0 + P.ncalls);
        if (active) {
            state[QUAD_ST_STAGE + QUAD_ST_M + s] = (uint64_t)__double_as_longlong(m);
            state[QUAD_ST_STAGE + QUAD_ST_V + s] = (uint64_t)(int64_t)r;
            state[QUAD_ST_STAGE + QUAD_ST_H + s] = hist;
        }
    }
}

__device__ __forceinline__ double *quad_stage_rot(const double2 *__restrict__ rot_cs, const cpm_quad_params &P)   // the caller synchronises
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *rot = reinterpret_cast<double *>(smem + P.rot_off);           // cos at [r], sin at [CPM_ROT_SIN + r]
    for (int k = threadIdx.x; k < 2 * P.p; k += blockDim.x) {
        const double2 e = rot_cs[k];
        rot[k] = e.x;
        rot[CPM_ROT_SIN + k] = e.y;
    }
    return rot;
}

template <int M_, int LP_>
#ifndef QUAD_MAIN_WAVES
#define QUAD_MAIN_WAVES 8
#endif
__global__ __launch_bounds__(QUAD_T, QUAD_MAIN_WAVES) void cpm_quad_kernel(const double2 *__restrict__ rows, const double2 *__restrict__ rot_cs,
                                                         uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                         uint64_t *__restrict__ edge, unsigned long long *__restrict__ unmerged,
                                                         cpm_quad_params P)
{
    const double *rot = quad_stage_rot(rot_cs, P);
    if (blockIdx.x == 0 && threadIdx.x < CPM_NLIST) cpm_list_counts(edge, P.nchunks, QUAD_EDGE_WORDS)[threadIdx.x] = 0;   // the repair lists: empty
    __syncthreads();
    cpm_quad_body<M_, LP_, false>(rows, rot, out, state, edge, unmerged, P, (int64_t)blockIdx.x, nullptr, nullptr);
}

// One repair round (list layout and invariant: wf_cpm_detect.h): a workgroup of EIGHT waves per listed chunk — team 0 from the
// chunk's recorded start, team 1 from the previous chunk's end.  finisher != 0: one workgroup that goes on, round after
// round, until a round hands nothing on.
template <int M_, int LP_>
__global__ __launch_bounds__(2 * QUAD_T) void cpm_quad_repair_kernel(const double2 *__restrict__ rows, const double2 *__restrict__ rot_cs,
                                                                    uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                                    uint64_t *__restrict__ edge, unsigned long long *__restrict__ unmerged,
                                                                    cpm_quad_params P, int lin, int lout, int finisher)
{
    uint64_t *const counts = cpm_list_counts(edge, P.nchunks, QUAD_EDGE_WORDS);
    int64_t n = (int64_t)__hip_atomic_load(&counts[lin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == 0) return;                                                   // (the whole grid)
    const double *rot = quad_stage_rot(rot_cs, P);
    __syncthreads();
    for (;;) {
        const uint64_t *list = cpm_list(edge, P.nchunks, QUAD_EDGE_WORDS, lin);
        for (int64_t idx = blockIdx.x; idx < n; idx += gridDim.x) {
            const uint64_t cw = __hip_atomic_load(&list[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int64_t chunk = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(cw >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)cw));
            cpm_quad_body<M_, LP_, true>(rows, rot, out, state, edge, unmerged, P, chunk, &counts[lout], cpm_list(edge, P.nchunks, QUAD_EDGE_WORDS, lout));
            __syncthreads();                                              // (the teams' buffers are reused)
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

// Every chunk against its predecessor: thread = (chunk c >= 1, state s); failed chunks are LISTED behind the records for
// the repair launches, or (repair = 0) counted as unproven.
__global__ __launch_bounds__(QUAD_T) void cpm_quad_verify_kernel(uint64_t *__restrict__ edge, int64_t nchunks, int S, uint64_t hmask,
                                                                 unsigned long long *__restrict__ unmerged, int repair)
{
    const int64_t c = (int64_t)blockIdx.x + 1;
    const int s = threadIdx.x;
    bool bad = false;
    if (c < nchunks && s < S) {
        const uint64_t *a = edge + c * QUAD_EDGE_WORDS, *b = edge + (c - 1) * QUAD_EDGE_WORDS + QUAD_EDGE_WORDS / 2;
        bad = a[3 * s] != b[3 * s] || a[3 * s + 1] != b[3 * s + 1] || ((a[3 * s + 2] ^ b[3 * s + 2]) & hmask) != 0ull;
    }
    const int any = __syncthreads_or(bad ? 1 : 0);
    if (s == 0 && any) {
        if (repair) {
            unsigned long long *counts = reinterpret_cast<unsigned long long *>(cpm_list_counts(edge, nchunks, QUAD_EDGE_WORDS));
            cpm_list(edge, nchunks, QUAD_EDGE_WORDS, 0)[atomicAdd(counts, 1ull)] = (uint64_t)c;
        } else {
            atomicAdd(unmerged, 1ull);
        }
    }
}

__global__ void cpm_quad_commit_kernel(uint64_t *state)
{
    for (int t = threadIdx.x; t < QUAD_ST_WORDS; t += blockDim.x) state[t] = state[QUAD_ST_STAGE + t];
}

static int quad_states(const wf_cpm_detector_config *d)
{
    int ncorr = 1;
    for (int i = 1; i < d->Lp; ++i) ncorr *= d->M;
    return d->NC * ncorr;
}

int wf_cpm_quad_applies(const wf_cpm_detector_config *d)
{
    if (!d || !(d->M == 2 || d->M == 4) || d->Lp < 2 || d->Lp > 3 || d->NC < 1) return 0;
    const int S = quad_states(d);
    return S > 64 && S <= 256;
}

// Default warm-up: 96 calls.  A chunk that misses it is run again from the true state until the two detectors meet, so the
// default is a matter of speed: 1e7 calls, ARTM 256 states, same box, steady state per block — 256 calls 11.74 - 11.92 ms and
// no repair at any Eb/N0; 128: 11.40 - 11.50 (0.5 repairs per block at 0 - 2 dB); 96: 11.35 - 11.48 (8 per block at 0 dB,
// none at 10); 64: 11.35 - 11.37 (13 per block already at 6 dB) — profiles/r06_ab_big_trellis_warmup.log.
int wf_cpm_quad_warmup(int warmup)                          // 0 = the default, a multiple of a batch, <= 4096
{
    int W = warmup ? warmup : 96;
    W = (W + 2 * QUAD_TB - 1) / (2 * QUAD_TB) * (2 * QUAD_TB);
    return W > 4096 ? 4096 : W;
}

// Chunk length: the smallest multiple of 64 that puts the burst into one round of resident detectors (8 workgroups per CU: 17.6 KB
// of LDS and 8 waves per SIMD), at least 512 and 2 W, at most 8192 (what one repair re-runs).
int64_t wf_cpm_quad_chunk_calls(int64_t ncalls, int W, int cus, int64_t chunk_opt)
{
    const int64_t slots = (int64_t)cus * 8;
    int64_t ch = ((ncalls + slots - 1) / slots + 63) / 64 * 64;
    if (ch < 512) ch = 512;
    if (ch < 2 * W) ch = (2 * W + 63) / 64 * 64;
    if (chunk_opt > 0) ch = (chunk_opt + 63) / 64 * 64;
    if (ch > 8192) ch = 8192;
    return ch;
}

int wf_cpm_quad_detect(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri, int64_t ncalls,
                       int warmup, uint8_t *d_decisions, void *d_state, void *stream, const cpm_mf_source *mf)
{
    cpm_quad_params P{};
    P.mf_templ = mf ? mf->d_templates : nullptr;
    P.mf_nsamp = mf ? mf->nsamp : 0;
    P.mf_start0 = mf ? mf->start0 : 0;
    P.mf_col0 = mf ? (mf->col0 & 1) : 0;
    WF_REQUIRE((det->M == 2 || det->M == 4) && det->Lp >= 2 && det->Lp <= 3 && (det->nh == 1 || det->nh == 2) && det->p >= 1 && det->p <= 64 &&
                   det->NC >= 1 && det->p % det->NC == 0 && det->D >= 1,
               "wf_cpm: unsupported detector (M %d Lp %d nh %d p %d NC %d D %d)", det->M, det->Lp, det->nh, det->p, det->NC, det->D);
    for (int i = 0; i < det->nh; ++i) WF_REQUIRE(det->K[i] >= 0 && det->K[i] < det->p, "wf_cpm: K[%d] = %d outside [0, p)", i, det->K[i]);
    const int lgM = det->M == 4 ? 2 : 1;
    WF_REQUIRE(det->D * lgM <= 64, "wf_cpm: decision delay %d does not fit the 64-bit decision register", det->D);
    P.M = det->M; P.p = det->p; P.nh = det->nh; P.K0 = det->K[0]; P.K1 = det->nh == 2 ? det->K[1] : det->K[0];
    P.Lp = det->Lp; P.NC = det->NC; P.D = det->D; P.S = quad_states(det);
    P.NF = 1;
    for (int i = 0; i < det->Lp; ++i) P.NF *= det->M;
    P.msub = det->Lp == 3 ? det->M : 1;
    WF_REQUIRE(P.S > 64 && P.S <= QUAD_T, "wf_cpm (quad form): %d states", P.S);
    const int W = wf_cpm_quad_warmup(warmup);
    P.CH = (int)wf_cpm_quad_chunk_calls(ncalls, W, ctx->cus, ctx->opt[WF_OPT_CPM_CHUNK_CALLS]);
    P.W = W;
    P.ncalls = ncalls;
    const int pieces = QUAD_TB * P.NF;
    P.rows_off = 0;
    P.xch_off = pieces * 16;
    P.src_off = P.xch_off + 4 * QUAD_XS * 8;
    P.min_off = P.src_off + QUAD_T * 4 + QUAD_T * 8;
    P.smp_off = P.min_off + 64;                                // MF form: 8 QUAD_TB + 1 samples (+ pad)
    P.dec_off = P.smp_off + (mf ? (8 * QUAD_TB + 2) * 16 : 0);  // (end of a team's block)
    P.team_bytes = (P.dec_off + 15) / 16 * 16;
    P.rot_off = 2 * P.team_bytes;                              // (the first launch uses one team's worth; the layout is the repair's)
    P.cmp_off = P.rot_off + 2 * CPM_ROT_SIN * 8;
    const size_t lds_repair = (size_t)P.cmp_off + 2 * 3 * QUAD_T * 8;
    cpm_quad_params P1 = P;                                    // first launch: one team, the rotation table right behind it
    P1.rot_off = P.team_bytes;
    const size_t lds_main = (size_t)P1.rot_off + 2 * CPM_ROT_SIN * 8;
    WF_REQUIRE(lds_repair <= 160 * 1024, "wf_cpm_viterbi_detect: chunk of %d calls does not fit LDS", P.CH);
    const int64_t nchunks = (ncalls + P.CH - 1) / P.CH;
    WF_REQUIRE(nchunks < (1ll << 31), "wf_cpm_viterbi_detect: burst too long for one launch");
    P.nchunks = P1.nchunks = nchunks;
    int rc = wf_ctx_reserve_vit(ctx, cpm_edge_total_words(nchunks, QUAD_EDGE_WORDS));
    if (rc) return rc;
    uint64_t *edge = reinterpret_cast<uint64_t *>(ctx->d_vit_edge);
    hipStream_t s = wf_stream(stream);
    using kern_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, unsigned long long *, cpm_quad_params);
    using repair_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, unsigned long long *, cpm_quad_params, int, int, int);
    kern_t k = nullptr;
    repair_t kr = nullptr;
    if (P.M == 4) {
        k = P.Lp == 2 ? cpm_quad_kernel<4, 2> : cpm_quad_kernel<4, 3>;
        kr = P.Lp == 2 ? cpm_quad_repair_kernel<4, 2> : cpm_quad_repair_kernel<4, 3>;
    } else {
        k = P.Lp == 2 ? cpm_quad_kernel<2, 2> : cpm_quad_kernel<2, 3>;
        kr = P.Lp == 2 ? cpm_quad_repair_kernel<2, 2> : cpm_quad_repair_kernel<2, 3>;
    }
    if (lds_main > 48 * 1024) WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_main));
    if (lds_repair > 48 * 1024) WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_repair));
    hipLaunchKernelGGL(k, dim3((unsigned)nchunks), dim3(QUAD_T), lds_main, s, reinterpret_cast<const double2 *>(d_rows_ri),
                       reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), edge, ctx->d_vit_unmerged, P1);
    WF_LAUNCH_CHECK();
    if (nchunks > 1) {
        const uint64_t hmask = lgM * P.D >= 64 ? ~0ull : ((1ull << (lgM * P.D)) - 1ull);
        const int repair = ctx->opt[WF_OPT_DET_REPAIR] == 0 ? 1 : 0;
        hipLaunchKernelGGL(cpm_quad_verify_kernel, dim3((unsigned)(nchunks - 1)), dim3(QUAD_T), 0, s, edge, nchunks, P.S, hmask, ctx->d_vit_unmerged, repair);
        WF_LAUNCH_CHECK();
        if (repair) {
            cpm_quad_params Pr = P;
            Pr.W = 0;
            for (int round = 0; round < 3; ++round) {           // two parallel rounds, then the finisher (wf_cpm_detect.h)
                hipLaunchKernelGGL(kr, dim3(round < 2 ? 2 * CPM_REPAIR_BLOCKS : 1), dim3(2 * QUAD_T), lds_repair, s, reinterpret_cast<const double2 *>(d_rows_ri),
                                   reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), edge,
                                   ctx->d_vit_unmerged, Pr, round, round + 1, round == 2 ? 1 : 0);
                WF_LAUNCH_CHECK();
            }
            if (ctx->opt[WF_OPT_DET_FINAL_VERIFY]) {
                hipLaunchKernelGGL(cpm_quad_verify_kernel, dim3((unsigned)(nchunks - 1)), dim3(QUAD_T), 0, s, edge, nchunks, P.S, hmask, ctx->d_vit_unmerged, 0);
                WF_LAUNCH_CHECK();
            }
        }
    }
    if (d_state) {
        hipLaunchKernelGGL(cpm_quad_commit_kernel, dim3(1), dim3(256), 0, s, static_cast<uint64_t *>(d_state));
        WF_LAUNCH_CHECK();
    }
    return WF_OK;
}
