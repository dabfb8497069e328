// wf_viterbi.hip — K8-K10: SOQPSK 4-state, 2-column Viterbi detector.
// Replaces SOQPSKTrellisDetector.iteration (reference waveforms/viterbi/algorithm.py:18-101):
// branch increments Re(c[start] * mf[idx(out)]) (:57-63), add-compare-select with MIN and a
// strict '<' (ties -> first branch in list order, :69-87), per-call min-normalisation of
// the carried metrics (:65-67), traceback of `length` stages from the first arg-min state
// (:90-98).
//
//  * viterbi_batch_kernel: length = 2 over a whole burst.  The reference recomputes a
//    2-stage window on every call; that equals a streaming ACS (stage 0 commits the
//    previous call's increments from the normalised carried metrics, stage 1 looks one
//    symbol ahead from the un-normalised result) + a depth-2 traceback, restated here
//    literally so that compare outcomes are bit-identical.  Parallelism: every thread owns
//    a chunk of consecutive calls and first re-derives the path metrics over `warmup`
//    earlier rows starting from zero metrics; once the survivors have merged (a few tens
//    of symbols on this 4-state trellis) the carried metric vector equals the sequential
//    one, hence so do all decisions.  The trellis (reference
//    waveforms/cpm/trellis/model.py:205-258) is compile-time: it is the only one the
//    reference detector supports (state_exp_term has 4 entries, algorithm.py:30).
//  * viterbi_iteration_kernel: one literal .iteration() for any window length, detector
//    state resident in device memory (drop-in for the per-symbol API).
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include "wf_common.h"

#define VIT_THREADS 256
// + 1 priming row: 32 rows = 8 batches of 4 before the first output row.  Measured over 4.4e6 chunks
// per warm-up length (tools/warmup_scan.py, Eb/N0 0 .. 12 dB, profiles/r02_warmup_scan.json): chunks
// that did not start from the sequential detector's metrics — 12 rows: 25 / 7 / 4 per 625 000 at
// 0 / 2 / 4 dB and none from 6 dB up; 16, 20, 24, 32, 48 rows: none at any Eb/N0.  Every launch still
// proves its own output (see the batch kernel), so this only sets how often a repair would run.
#define VIT_DEFAULT_WARMUP 31
#define VIT_MAX_LEN 64

// Branch b of column c: start = b >> 1; ends / output-symbol index (0: -2, 1: 0, 2: +2):
//   column 0 (even / I): end = (start & 1) + 2*(b & 1)
//   column 1 (odd  / Q): end = (start & 2) + (b & 1)
// (model.py:205-230; the diff-encoded trellis :233-258 only relabels the inputs).
// {{1, 2, 1, 0, 0, 1, 2, 1}, {1, 0, 2, 1, 1, 2, 0, 1}} as 2-bit fields of one literal: a table in constant memory
// is a dependent ~0.5 us load each time the per-symbol server (one wave on an idle chip) looks an entry up
__device__ __forceinline__ int br_out_idx(int col, int b) { return (int)((0x49616419u >> (2 * (8 * col + b))) & 3u); }

__device__ __forceinline__ int br_end(int col, int b)
{
    const int s = b >> 1;
    return col == 0 ? (s & 1) + 2 * (b & 1) : (s & 2) + (b & 1);
}

__device__ __forceinline__ int br_inp(int col, int b, int diff)
{
    const int s = b >> 1;
    const int flip = diff ? (col == 0 ? (s >> 1) : (s & 1)) : 0;
    return (b & 1) ^ flip;
}

// Re(state_exp_term[start] * z), state_exp_term = [+1j, -1, +1, -1j] (algorithm.py:30)
__device__ __forceinline__ double br_inc(int start, double re, double im)
{
    return start == 0 ? -im : start == 1 ? -re : start == 2 ? re : im;
}

// ---- lean detector step for the batch kernel --------------------------------------------
// Per trellis section only 4 of the 6 components of a matched-filter row are ever used:
// Re/Im of z[1] (alpha = 0) in both sections, plus (Re z[0], Im z[2]) in the even (I) section
// and (Im z[0], Re z[2]) in the odd (Q) one.  Incoming branches per end state, first in
// LIST order (ties keep the first: strict '<', algorithm.py:79-83), with their increment
// Re(state_exp_term[start] * z[idx(out)]) written as a signed component:
//   even: st0: (0: -i1 | 2: +a)   st1: (1: -r1 | 3: +b)   st2: (0: -b | 2: +r1)   st3: (1: -a | 3: +i1)
//   odd : st0: (0: -i1 | 1: -b)   st1: (0: -a | 1: -r1)   st2: (2: +r1 | 3: +a)   st3: (2: +b | 3: +i1)
// with r1 = Re z[1], i1 = Im z[1], (a, b) = (Re z[0], Im z[2]) even / (Im z[0], Re z[2]) odd.
// m + (-x) and m - x are the same IEEE operation, so every compare is bit-identical to the
// reference's.
struct vit_comp {
    double r1, i1, a, b;
};

template <int COL, bool PACKED>
__device__ __forceinline__ vit_comp vit_components(const double2 *__restrict__ z)
{
    vit_comp c;
    if (PACKED) {   // row = {r1, i1, a, b}: the bank already picked the section's components
        c.r1 = z[0].x;
        c.i1 = z[0].y;
        c.a = z[1].x;
        c.b = z[1].y;
        return c;
    }
    c.r1 = z[1].x;
    c.i1 = z[1].y;
    c.a = COL == 0 ? z[0].x : z[0].y;
    c.b = COL == 0 ? z[2].y : z[2].x;
    return c;
}

// one ACS stage of section COL: second[s] = the second listed branch won state s.
// The survivor metric is min(fa, fb): identical to the reference's `fb < fa ? fb : fa` for every
// ordered pair (equal values differ at most in the sign of a zero, which no later compare or
// sum can see), and one v_min_f64 instead of a compare and two 32-bit selects.  The selection
// flags are WAVE MASKS (the ballot of the compare is the SGPR pair v_cmp writes anyway), so the
// whole traceback below is scalar mask logic (s_and_b64 / s_or_b64 / s_xor_b64) that costs the
// vector pipe nothing; a lane reads its own bit back with one v_cndmask (inverse ballot).
template <int COL>
__device__ __forceinline__ void vit_acs(const double m[4], const vit_comp &q, double out[4], uint64_t second[4])
{
    double fa[4], fb[4];
    if (COL == 0) {
        fa[0] = m[0] - q.i1; fb[0] = m[2] + q.a;
        fa[1] = m[1] - q.r1; fb[1] = m[3] + q.b;
        fa[2] = m[0] - q.b;  fb[2] = m[2] + q.r1;
        fa[3] = m[1] - q.a;  fb[3] = m[3] + q.i1;
    } else {
        fa[0] = m[0] - q.i1; fb[0] = m[1] - q.b;
        fa[1] = m[0] - q.a;  fb[1] = m[1] - q.r1;
        fa[2] = m[2] + q.r1; fb[2] = m[3] + q.a;
        fa[3] = m[2] + q.b;  fb[3] = m[3] + q.i1;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        second[s] = __builtin_amdgcn_ballot_w64(fb[s] < fa[s]);
        out[s] = fmin(fa[s], fb[s]);
    }
}

__device__ __forceinline__ uint64_t vit_mux(uint64_t c, uint64_t x, uint64_t y) { return (c & x) | (~c & y); }

struct vit_lane {
    double m0[4];
    vit_comp prev;                       // components of the previous call's row
    uint32_t wb, ws;                     // decisions of the current group of 16 calls: 1 bit / 2 bits each
    double ms[4];                        // metrics the warm-up arrived at, just before the lane's first own call
};

// One detector call (COL = column parity of the call): stage 0 commits the previous call's
// increments from the min-normalised carried metrics, stage 1 looks one symbol ahead,
// depth-2 traceback from the first arg-min (algorithm.py:57-101 with length = 2).
// Decision out: bit = input bit, cd0 | cd1 << 1 = sym / 2 + 1.
template <int COL, bool PACKED>
__device__ __forceinline__ void vit_step_core(vit_lane &L, const double2 *__restrict__ zrow, int diff, bool &bit, bool &cd0, bool &cd1)
{
    constexpr int PREV = COL ^ 1;
    const vit_comp now = vit_components<COL, PACKED>(zrow);
    const double mn = fmin(fmin(L.m0[0], L.m0[1]), fmin(L.m0[2], L.m0[3]));
    const double carried[4] = {L.m0[0] - mn, L.m0[1] - mn, L.m0[2] - mn, L.m0[3] - mn};
    double ma[4], mb[4];
    uint64_t n0[4], n1[4];
    vit_acs<PREV>(carried, L.prev, ma, n0);   // stage j = 0
    vit_acs<COL>(ma, now, mb, n1);            // stage j = 1
    // np.argmin (first minimum) and the depth-2 traceback, as lane-mask logic.
    //   s1 = arg-min state of stage 1 (bits s1b1 s1b0), w1 = which branch entered it
    const uint64_t c01 = __builtin_amdgcn_ballot_w64(mb[1] < mb[0]), c23 = __builtin_amdgcn_ballot_w64(mb[3] < mb[2]);
    const double v01 = fmin(mb[0], mb[1]), v23 = fmin(mb[2], mb[3]);
    const uint64_t s1b1 = __builtin_amdgcn_ballot_w64(v23 < v01);
    const uint64_t s1b0 = vit_mux(s1b1, c23, c01);
    const uint64_t w1 = vit_mux(s1b1, vit_mux(c23, n1[3], n1[2]), vit_mux(c01, n1[1], n1[0]));
    //   e0 = state after stage 0 = predecessor of s1 in section COL:
    //        even (st & 1) + 2 * second, odd (st & 2) + second
    const uint64_t e0b0 = COL == 0 ? s1b0 : w1;
    const uint64_t e0b1 = COL == 0 ? w1 : s1b1;
    const uint64_t w0 = vit_mux(e0b1, vit_mux(e0b0, n0[3], n0[2]), vit_mux(e0b0, n0[1], n0[0]));   // which branch entered e0
    //   st0 = start state of that branch (predecessor of e0 in section PREV); branch index
    //   b = 2 * st0 + (PREV == 0 ? e0 >> 1 : e0 & 1)
    const uint64_t st0b0 = PREV == 0 ? e0b0 : w0;
    const uint64_t st0b1 = PREV == 0 ? w0 : e0b1;
    const uint64_t b0 = PREV == 0 ? e0b1 : e0b0, b1 = st0b0, b2 = st0b1;
    //   input bit (br_inp): (b & 1) ^ (diff ? (PREV == 0 ? st0 >> 1 : st0 & 1) : 0)
    const uint64_t bit_m = b0 ^ (diff ? (PREV == 0 ? st0b1 : st0b0) : 0ull);
    //   output symbol by branch (model.py:205-258): even {0,+2,0,-2,-2,0,+2,0}, odd {0,-2,+2,0,0,+2,-2,0};
    //   code = sym / 2 + 1: even  code&1 <=> b0 == b2, code&2 <=> b0 != b2 && b1 == b2
    //                       odd   code&1 <=> b0 == b1, code&2 <=> b0 != b1 && b2 == b0
    const uint64_t x02 = b0 ^ b2, x01 = b0 ^ b1, x12 = b1 ^ b2;
    const uint64_t cd0_m = PREV == 0 ? ~x02 : ~x01;
    const uint64_t cd1_m = PREV == 0 ? (x02 & ~x12) : (x01 & ~x02);
    bit = __builtin_amdgcn_inverse_ballot_w64(bit_m);
    cd0 = __builtin_amdgcn_inverse_ballot_w64(cd0_m);
    cd1 = __builtin_amdgcn_inverse_ballot_w64(cd1_m);
#pragma unroll
    for (int s = 0; s < 4; ++s) L.m0[s] = ma[s];
    L.prev = now;
}

// ... of the batch kernel: call k of a lane that owns calls [a, ...); prime = the row before the warm-up (its exact
// components only).
template <int COL, bool PACKED>
__device__ __forceinline__ void vit_step(vit_lane &L, const double2 *__restrict__ zrow, bool prime, int diff,
                                         int64_t k, int64_t a, int64_t ncalls, uint64_t *__restrict__ dec)
{
    if (prime) {   // priming row: exact components of the call before the warm-up
        L.prev = vit_components<COL, PACKED>(zrow);
        return;
    }
    bool bit, cd0, cd1;
    vit_step_core<COL, PACKED>(L, zrow, diff, bit, cd0, cd1);
    if (k >= a) {
        // Decisions stay on chip until the lane's chunk is done: packed here (bit c: input bit;
        // bits 2c..2c+1: sym / 2 + 1), one 8 B word per 16 calls parked in the wave's LDS strip,
        // expanded and stored by vit_flush after the last row has been loaded.  A global store in
        // this loop would share vmcnt with the row prefetch and force every wait down to
        // vmcnt(0) (mixed loads and stores complete out of order on gfx9-family counters).
        const int c = (int)(k - a) & 15;
        L.wb |= (bit ? 1u : 0u) << c;
        L.ws |= ((cd0 ? 1u : 0u) | (cd1 ? 2u : 0u)) << (2 * c);
        if (c == 15 || k + 1 == ncalls) {
            dec[(int)(k - a) >> 4] = (uint64_t)L.wb | ((uint64_t)L.ws << 32);
            L.wb = L.ws = 0;
        }
    }
}

// Expand the packed decisions of one lane (calls [a, a + n)) and store them: 16 B per group of 16.
__device__ __forceinline__ uint64_t vit_spread_bits8(uint32_t x)    // 8 bits -> 8 bytes of 0 / 1, LSB first
{
    const uint64_t t = ((uint64_t)(x & 0xFF) * 0x0101010101010101ull) & 0x8040201008040201ull;
    return ((t + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
}

__device__ __forceinline__ uint64_t vit_spread_syms8(uint32_t x)    // 8 two-bit codes -> 8 bytes of -2 / 0 / +2
{
    uint64_t t = ((uint64_t)(x & 0xFFFF) | ((uint64_t)(x & 0xFFFF) << 24)) & 0x000000FF000000FFull;
    t = (t | (t << 12)) & 0x000F000F000F000Full;
    t = (t | (t << 6)) & 0x0303030303030303ull;
    const uint64_t v = t << 1;                                       // 0, 2, 4 per byte; + 0xFE (mod 256) = -2, 0, +2
    return ((v & 0x7F7F7F7F7F7F7F7Full) + 0x7E7E7E7E7E7E7E7Eull) ^ ((v ^ 0xFEFEFEFEFEFEFEFEull) & 0x8080808080808080ull);
}

__device__ __forceinline__ void vit_flush(const uint64_t *__restrict__ dec, int ch, int64_t a, int64_t ncalls,
                                          uint8_t *__restrict__ bits, int8_t *__restrict__ syms)
{
    for (int gq = 0; gq < ch / 16; ++gq) {
        const int64_t k0 = a + 16 * gq;
        if (k0 >= ncalls) break;
        const uint64_t w = dec[gq];
        const uint32_t wb = (uint32_t)w, ws = (uint32_t)(w >> 32);
        const uint64_t b_lo = vit_spread_bits8(wb), b_hi = vit_spread_bits8(wb >> 8);
        const uint64_t s_lo = vit_spread_syms8(ws), s_hi = vit_spread_syms8(ws >> 16);
        if (k0 + 16 <= ncalls) {
            *reinterpret_cast<ulonglong2 *>(bits + k0) = make_ulonglong2(b_lo, b_hi);
            *reinterpret_cast<ulonglong2 *>(syms + k0) = make_ulonglong2(s_lo, s_hi);
        } else {   // ragged tail of the burst
            for (int q = 0; k0 + q < ncalls; ++q) {
                bits[k0 + q] = (uint8_t)(((q < 8 ? b_lo : b_hi) >> (8 * (q & 7))) & 0xFF);
                syms[k0 + q] = (int8_t)(((q < 8 ? s_lo : s_hi) >> (8 * (q & 7))) & 0xFF);
            }
        }
    }
}

__device__ __forceinline__ void vit_save_start(vit_lane &L)
{
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) L.ms[s4] = L.m0[s4];
}

// Carry block <-> lane state.  The block keeps the reference's shape (8 branch increments of
// the last call); section `col` of that call tells which components they hold.
__device__ __forceinline__ vit_comp vit_comp_from_inc(const double *inc, int col)
{
    vit_comp c;
    c.i1 = inc[7];
    if (col == 0) { c.a = inc[4]; c.r1 = inc[5]; c.b = inc[6]; }
    else { c.r1 = inc[4]; c.b = inc[5]; c.a = inc[6]; }
    return c;
}

__device__ __forceinline__ void vit_comp_to_inc(const vit_comp &c, int col, double *inc)
{
    if (col == 0) {
        inc[0] = -c.i1; inc[1] = -c.b; inc[2] = -c.r1; inc[3] = -c.a;
        inc[4] = c.a;   inc[5] = c.r1; inc[6] = c.b;   inc[7] = c.i1;
    } else {
        inc[0] = -c.i1; inc[1] = -c.a; inc[2] = -c.b;  inc[3] = -c.r1;
        inc[4] = c.r1;  inc[5] = c.b;  inc[6] = c.a;   inc[7] = c.i1;
    }
}

// Batch kernel.  Lane g owns calls [g*CH, (g+1)*CH) and walks rows g*CH - W - 1 ... (one
// priming row for the exact previous increments, W warm-up rows, CH output rows).  Rows are
// 48 B each (32 B when the bank packed them) and a lane's rows are contiguous, so per-lane
// loads would touch 64 different cache lines per instruction and thrash L1 (measured 2.4x HBM
// over-fetch).  Instead each wave stages VIT_S rows of all its 64 lanes per batch with
// COOPERATIVE loads — consecutive lanes cover one lane-segment of VIT_S rows, every 16 B piece
// of every cache line is fetched once — into a wave-private LDS tile (an odd number of 16 B
// slots per lane makes the per-lane ds_read_b128 conflict-free), with VIT_DEPTH batches of
// global loads in flight while the current one is being decoded.  Steps are decoded in
// (even, odd) column pairs so only two ACS bodies are live at a time.
#define VIT_S 4
#ifndef VIT_DEPTH
#define VIT_DEPTH 3   // batches of cooperative loads in flight per wave (register sets pend0..2); 2 or 3
#endif
#define VIT_PIECES_RW(RW) ((RW) * VIT_S)   // 16-byte pieces per lane-segment, RW = double2 per row (3, packed: 2)
#define VIT_PIECES (3 * VIT_S)
#define VIT_LANE_SLOTS (VIT_PIECES + 1)     // LDS slots per lane as allocated (odd count: conflict-free ds_read_b128)

// 16 B load as a VALUE: assigning `dst[u] = ptr[i]` for the double2 class type lowers to a memcpy
// into the private array, which then stays in scratch memory instead of registers.
__device__ __forceinline__ double2 vit_ld16(const double2 *p)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));   // error counter reads them (same-box: detector 85.6 -> 86.3 us, counter 17.3 -> 15.9)
    return make_double2(v.x, v.y);
}

template <int PAR0, bool PACKED>   // PAR0: column parity of step 0's call index; PACKED: 32 B rows
__device__ __forceinline__ void viterbi_batch_body(const double *__restrict__ mf, int64_t ncalls, int CH, int diff, int warmup,
                                                   uint8_t *__restrict__ bits, int8_t *__restrict__ syms,
                                                   double *__restrict__ state, int64_t i0, double2 (*s_rows)[WF_WAVE * VIT_LANE_SLOTS],
                                                   uint64_t *__restrict__ s_dec, double *__restrict__ edge)
{
    constexpr int RW = PACKED ? 2 : 3;                 // double2 per row
    constexpr int NP = VIT_PIECES_RW(RW);             // 16-byte pieces per lane-segment and batch
    constexpr int LS = NP + 1;                         // LDS slots per lane (odd: conflict-free)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g0 = ((int64_t)blockIdx.x * (VIT_THREADS / WF_WAVE) + wave) * WF_WAVE;  // first lane of the wave
    const int64_t a = (g0 + lane) * CH;
    const bool live = a < ncalls;
    const int nsteps = warmup + 1 + CH;
    const int nbatch = (nsteps + VIT_S - 1) / VIT_S;
    const double2 *rows = reinterpret_cast<const double2 *>(mf);
    double2 *tile = s_rows[wave];
    uint64_t *dec = s_dec + (wave * WF_WAVE + lane) * (CH / 16 + 1);   // + 1: odd 8 B stride, conflict-free

    // cooperative piece p = u*64 + lane  ->  (segment = lane of the wave it belongs to, piece index).
    // Everything per-lane is 32-bit and batch-invariant; the batch only moves a wave-uniform base.
    // (segment, piece) of this lane's u-th cooperative piece, packed; LDS slot, row and global
    // offset are derived from it where needed (keeps 24 registers out of the loop)
    int sgw[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int p = u * WF_WAVE + lane;
        const int sg = p / NP;
        sgw[u] = (sg << 8) | (p - sg * NP);
    }
    const int64_t g0u = __builtin_amdgcn_readfirstlane((int)(g0 >> 6)) * (int64_t)WF_WAVE;   // wave-uniform copy of g0
    // Rows outside [0, ncalls) exist only around the first and last chunk of a burst and are never
    // decoded (vit_step is skipped for them), so their loads are merely redirected to a valid
    // address: a wave-uniform test picks the unclamped form for every interior batch.
    auto fetch = [&](int b, double2 dst[NP]) __attribute__((always_inline)) {
        const int64_t rb = g0u * CH - warmup - 1 + (int64_t)b * VIT_S;   // first row of the batch (uniform)
        const double2 *basep = rows + RW * rb;
        if (rb >= 0 && rb + (int64_t)(WF_WAVE - 1) * CH + VIT_S <= ncalls) {
#pragma unroll
            for (int u = 0; u < NP; ++u) dst[u] = vit_ld16(basep + RW * CH * (sgw[u] >> 8) + (sgw[u] & 255));
        } else {
            const int64_t lo64 = -rb, hi64 = ncalls - 1 - rb;             // valid rrel range [lo, hi]
            const int lo = lo64 < -(1 << 30) ? -(1 << 30) : (lo64 > (1 << 30) ? (1 << 30) : (int)lo64);
            const int hi = hi64 < -(1 << 30) ? -(1 << 30) : (hi64 > (1 << 30) ? (1 << 30) : (int)hi64);
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int sg = sgw[u] >> 8, wi = sgw[u] & 255;
                const int rrel = sg * CH + wi / RW;     // row relative to the wave's first row of the batch
                const int rc = min(max(rrel, lo), hi);
                dst[u] = vit_ld16(basep + RW * rc + (wi - RW * (wi / RW)));
            }
        }
    };
    auto stash = [&](const double2 src[NP]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NP; ++u) tile[(sgw[u] >> 8) * LS + (sgw[u] & 255)] = src[u];
    };

    vit_lane L;
#pragma unroll
    for (int k = 0; k < 4; ++k) L.m0[k] = 0.0;
    L.prev.r1 = L.prev.i1 = L.prev.a = L.prev.b = 0.0;
    L.wb = L.ws = 0;
    if (state && a == 0) {   // the lane that starts the burst continues the carried detector
#pragma unroll
        for (int k = 0; k < 4; ++k) L.m0[k] = state[1 + k];
        L.prev = vit_comp_from_inc(state + 5, (int)((i0 - 1) & 1));
    }
    // The tile is wave-private and LDS executes a wave's accesses in order, so the loop needs no
    // workgroup barrier at all: only the loads gate it.  One batch of lookahead left a wave
    // waiting a full memory round trip (~4 us under load) per 4 decoded steps, with ~1 wave per
    // SIMD resident -> 207 us for 176 steps.  VIT_DEPTH batches are kept in flight in registers
    // instead (VIT_DEPTH x 12 x 16 B per lane; the register file is nearly empty at this occupancy).
#if VIT_DEPTH == 3
    double2 pend0[NP], pend1[NP], pend2[NP];
#else
    double2 pend0[NP], pend1[NP];
#endif
    // Every round issues exactly one batch of loads, unconditionally (past the last batch it
    // re-reads that batch: L2 hits, results unused).  With a conditional refill the compiler must
    // assume the path on which nothing was issued behind a register set and waits with vmcnt(0).
    const int last_b = nbatch - 1;
    fetch(0, pend0);
    fetch(min(1, last_b), pend1);
#if VIT_DEPTH == 3
    fetch(min(2, last_b), pend2);
#endif
    const int64_t kbase = a - warmup - 1;
    auto round = [&](int b, double2 (&pd)[NP]) __attribute__((always_inline)) {
        stash(pd);                                           // waits for batch b only (loads return in order)
        fetch(min(b + VIT_DEPTH, last_b), pd);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (live) {
#pragma unroll 1
            for (int jj = 0; jj < VIT_S; jj += 2) {
                const int j = b * VIT_S + jj;
                const int64_t k = kbase + j;                  // call index of the even step
                const double2 *zr = tile + lane * LS + RW * jj;
                if (j == warmup + 1) vit_save_start(L);       // wave-uniform: the next step is call a
                if (j < nsteps && k >= 0 && k < ncalls)
                    vit_step<PAR0, PACKED>(L, zr, j == 0, diff, k, a, ncalls, dec);
                if (j + 1 == warmup + 1) vit_save_start(L);
                if (j + 1 < nsteps && k + 1 >= 0 && k + 1 < ncalls)
                    vit_step<PAR0 ^ 1, PACKED>(L, zr + RW, false, diff, k + 1, a, ncalls, dec);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                     // batch b consumed before the next stash
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int b0 = 0; b0 < nbatch; b0 += VIT_DEPTH) {   // rounds past nbatch decode nothing (j >= nsteps)
        round(b0, pend0);
        round(b0 + 1, pend1);
#if VIT_DEPTH == 3
        round(b0 + 2, pend2);
#endif
    }
    if (live) vit_flush(dec, CH, a, ncalls, bits, syms);   // same lane wrote the strip: in-order LDS, no barrier
    // Proof obligation of the chunk-parallel form: the metrics a lane's warm-up arrived at must be
    // BITWISE the metrics its predecessor ended with (then every later compare is the sequential
    // detector's).  Every lane records what its own calls started from and what they ended with
    // (64 B per chunk, one coalesced 4 KB run per wave); viterbi_fixup_kernel compares neighbours and
    // runs the chunks that differ again from the true metrics.
    if (edge && live) {
        double *e = edge + 8 * (g0 + lane);
        *reinterpret_cast<double4 *>(e) = make_double4(L.ms[0], L.ms[1], L.ms[2], L.ms[3]);
        *reinterpret_cast<double4 *>(e + 4) = make_double4(L.m0[0], L.m0[1], L.m0[2], L.m0[3]);
    }
    if (state && live && a + CH >= ncalls) {
        // the lane that owns the last call hands the detector state on (streaming).  Written
        // to the second half of the carry block; viterbi_carry_commit_kernel moves it.
        state[16] = (double)(i0 + ncalls);
#pragma unroll
        for (int k = 0; k < 4; ++k) state[17 + k] = L.m0[k];
        vit_comp_to_inc(L.prev, (int)((i0 + ncalls - 1) & 1), state + 21);
    }
}

#ifndef VIT_MIN_WAVES
#define VIT_MIN_WAVES 1
#endif
// Proof records and repair lists of the 4-state detectors (batch and window form alike), in the context's scratch:
//   rec[nchunks][8]   doubles: {the C / metrics a chunk's own calls started from [4], what they ended with [4]}
//   hdr[VIT_HDR]      u64: [0], [1] entries in list 0 / 1, [2] the fix-up launch's arrival ticket
//   list[2][nchunks]  u64 chunk indices
// viterbi_fixup_kernel / vwin_fixup_kernel, ONE launch behind the detector: every workgroup compares its share of the
// chunk boundaries and lists the chunks whose start is not bitwise their predecessor's end; the workgroup that arrives
// last then repairs: a thread per listed chunk runs the chunk's calls AGAIN from the predecessor's end (which becomes the
// chunk's recorded start), rewrites its decisions and its end, and lists the next chunk when that end changed — round
// after round (lists 0 <-> 1) until a round lists nothing.  Every round's smallest chunk is run from the true state, so
// the consistent prefix grows each round: at worst the rounds are the sequential detector (algorithm.py:44-101), and
// the warm-up only decides how often any of this runs (at the default, at any Eb/N0 measured: never).
#define VIT_HDR 8
__host__ __device__ inline size_t vit_edge_words(int64_t nchunks) { return (size_t)nchunks * 8 + VIT_HDR + 2 * (size_t)nchunks; }

template <bool PACKED>
__global__ __launch_bounds__(VIT_THREADS, VIT_MIN_WAVES) void viterbi_batch_kernel(const double *__restrict__ mf, int64_t ncalls,
                                                                     int ch, int diff, int warmup, uint8_t *__restrict__ bits,
                                                                     int8_t *__restrict__ syms, double *__restrict__ state,
                                                                     double *__restrict__ edge, int64_t nchunks)
{
    __shared__ double2 s_rows[VIT_THREADS / WF_WAVE][WF_WAVE * VIT_LANE_SLOTS];
    extern __shared__ uint64_t s_dec[];   // packed decisions, one strip of ch / 16 + 1 words per lane
    const int64_t i0 = state ? (int64_t)state[0] : 0;
    if (blockIdx.x == 0 && threadIdx.x < VIT_HDR) reinterpret_cast<uint64_t *>(edge + 8 * nchunks)[threadIdx.x] = 0;   // lists empty, nobody arrived
    // call index of step 0 is lane_start - warmup - 1 with lane_start a multiple of ch (even)
    if ((i0 - warmup - 1) & 1) viterbi_batch_body<1, PACKED>(mf, ncalls, ch, diff, warmup, bits, syms, state, i0, s_rows, s_dec, edge);
    else viterbi_batch_body<0, PACKED>(mf, ncalls, ch, diff, warmup, bits, syms, state, i0, s_rows, s_dec, edge);
}

// The part of the fix-up launch both forms share: compare, list (or, mode 0, count), and elect the last workgroup.
// Returns true in every thread of the workgroup that arrived last (all others are done).  mode: 1 = list for repair,
// 0 = count as unproven (WF_OPT_DET_REPAIR off; the closing check of WF_OPT_DET_FINAL_VERIFY).
__device__ __forceinline__ bool vit_fixup_verify(double *__restrict__ edge, int64_t nchunks, unsigned long long *__restrict__ unmerged, int mode)
{
    __shared__ int s_last;
    unsigned long long *hdr = reinterpret_cast<unsigned long long *>(edge + 8 * nchunks);
    unsigned long long *list0 = hdr + VIT_HDR;
    const unsigned long long *rec = reinterpret_cast<const unsigned long long *>(edge);
    int listed = 0;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; c < nchunks; c += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long *st = rec + 8 * c, *en = rec + 8 * (c - 1) + 4;
        if (st[0] != en[0] || st[1] != en[1] || st[2] != en[2] || st[3] != en[3]) {
            if (mode) list0[atomicAdd(&hdr[0], 1ull)] = (unsigned long long)c;
            else atomicAdd(unmerged, 1ull);
            listed = 1;
        }
    }
    if (!mode) return false;
    // Arrival.  A workgroup that listed something publishes its entries (agent-scope release) before it arrives; the
    // others have nothing to publish and only arrive — in the usual launch nobody lists anything, and a release fence
    // writes back the XCD's whole L2 under the front end that is streaming rows through it on the other stream.  The
    // list COUNT needs no fence: it and the ticket are device-scope atomics, and every thread's count increment has
    // returned (the barrier waits for it) before thread 0 takes the ticket.
    if (__syncthreads_or(listed)) __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&hdr[2], 1ull) + 1 == (unsigned long long)gridDim.x;
    __syncthreads();
    if (!s_last) return false;
    if (__hip_atomic_load(&hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return false;   // nothing to repair: the usual case
    __threadfence();                                       // ... everybody's entries before the repairs read them
    return true;
}

// The round loop of the last workgroup: `repair(c)` runs chunk c again and returns true when its end changed.
template <class F>
__device__ __forceinline__ void vit_fixup_rounds(double *__restrict__ edge, int64_t nchunks, unsigned long long *__restrict__ unmerged, F &&repair)
{
    unsigned long long *hdr = reinterpret_cast<unsigned long long *>(edge + 8 * nchunks);
    int lin = 0;
    for (;;) {
        const int64_t n = (int64_t)__hip_atomic_load(&hdr[lin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == 0) return;
        unsigned long long *in = hdr + VIT_HDR + (int64_t)lin * nchunks, *out = hdr + VIT_HDR + (int64_t)(lin ^ 1) * nchunks;
        for (int64_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
            const int64_t c = (int64_t)__hip_atomic_load(&in[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool changed = repair(c);
            atomicAdd(unmerged + 1, 1ull);                 // [1]: chunk repairs run, [2]: ... whose end changed (handed on)
            if (changed) {
                atomicAdd(unmerged + 2, 1ull);
                if (c + 1 < nchunks) out[atomicAdd(&hdr[lin ^ 1], 1ull)] = (unsigned long long)(c + 1);
            }
        }
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&hdr[lin], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: the round after next fills it
        __threadfence();
        __syncthreads();
        lin ^= 1;
    }
}

// Chunk c of the length-2 detector again, by ONE thread straight from global memory: metrics from the predecessor's
// end, the previous call's components from its own row (exact), every call of the chunk, decisions rewritten.
template <int PAR, bool PACKED>                            // PAR: column parity of the chunk's first call (chunks start on even local calls)
__device__ __forceinline__ bool vit_rerun_chunk(const double *__restrict__ mf, int64_t ncalls, int CH, int diff, uint8_t *__restrict__ bits,
                                                int8_t *__restrict__ syms, double *__restrict__ state, int64_t i0, double *__restrict__ edge, int64_t c)
{
    constexpr int RW = PACKED ? 2 : 3;
    const double2 *rows = reinterpret_cast<const double2 *>(mf);
    const int64_t a = c * CH, kend = a + CH < ncalls ? a + CH : ncalls;
    double *rec = edge + 8 * c;
    vit_lane L;
    L.wb = L.ws = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double v = __hip_atomic_load(rec - 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the predecessor's end, as it is now
        L.m0[q] = v;
        rec[q] = v;                                        // ... is what this run of the chunk starts from
    }
    L.prev = vit_components<PAR ^ 1, PACKED>(rows + RW * (a - 1));      // (a >= CH: only chunks c >= 1 are ever listed)
    for (int64_t k = a; k < kend; k += 2) {
        bool bit, cd0, cd1;
        vit_step_core<PAR, PACKED>(L, rows + RW * k, diff, bit, cd0, cd1);
        bits[k] = bit ? 1 : 0;
        syms[k] = (int8_t)(2 * ((cd0 ? 1 : 0) | (cd1 ? 2 : 0)) - 2);
        if (k + 1 < kend) {
            vit_step_core<PAR ^ 1, PACKED>(L, rows + RW * (k + 1), diff, bit, cd0, cd1);
            bits[k + 1] = bit ? 1 : 0;
            syms[k + 1] = (int8_t)(2 * ((cd0 ? 1 : 0) | (cd1 ? 2 : 0)) - 2);
        }
    }
    bool changed = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        changed |= __double_as_longlong(rec[4 + q]) != __double_as_longlong(L.m0[q]);
        rec[4 + q] = L.m0[q];
    }
    if (changed && state && a + CH >= ncalls) {            // the chunk that owns the last call: the carry (staging half, see viterbi_batch_body)
#pragma unroll
        for (int q = 0; q < 4; ++q) state[17 + q] = L.m0[q];
    }
    return changed;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void viterbi_fixup_kernel(const double *__restrict__ mf, int64_t ncalls, int ch, int diff,
                                                            uint8_t *__restrict__ bits, int8_t *__restrict__ syms, double *__restrict__ state,
                                                            double *__restrict__ edge, int64_t nchunks, unsigned long long *__restrict__ unmerged, int mode)
{
    if (!vit_fixup_verify(edge, nchunks, unmerged, mode)) return;
    const int64_t i0 = state ? (int64_t)state[0] : 0;
    if (i0 & 1)
        vit_fixup_rounds(edge, nchunks, unmerged, [&](int64_t c) { return vit_rerun_chunk<1, PACKED>(mf, ncalls, ch, diff, bits, syms, state, i0, edge, c); });
    else
        vit_fixup_rounds(edge, nchunks, unmerged, [&](int64_t c) { return vit_rerun_chunk<0, PACKED>(mf, ncalls, ch, diff, bits, syms, state, i0, edge, c); });
}

__global__ void viterbi_carry_commit_kernel(double *state)
{
    const int t = threadIdx.x;
    if (t < 16) state[t] = state[16 + t];
}

static int viterbi_launch(wf_ctx *ctx, const double *d_mf_ri, int64_t ncalls, int differential, int warmup,
                          uint8_t *d_bits, int8_t *d_syms, double *d_state, void *stream, bool packed = false)
{
    WF_REQUIRE(ctx && ncalls >= 0 && warmup >= 0, "wf_viterbi4_detect: bad argument");
    if (ncalls == 0) return WF_OK;
    WF_REQUIRE(d_mf_ri && d_bits && d_syms, "wf_viterbi4_detect: NULL device pointer");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_mf_ri) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_bits) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_syms) & 15) == 0,
               "wf_viterbi4_detect: device pointers must be 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    if (warmup == 0) warmup = VIT_DEFAULT_WARMUP;
    if (warmup > 4096) warmup = 4096;
    // Calls per lane (a multiple of 16).  The kernel holds 3 batches of row loads in registers and
    // runs ONE wave per SIMD, so a launch of more than 256 workgroups (one per CU) would need a
    // second round: the chunk is the smallest that fits the burst into 256 workgroups — every lane
    // walks ch + 48 dependent steps, so smaller is faster, and the warm-up re-reads 48 / ch of the
    // rows.  (1e7 calls: ch = 160, 0.112 ms; at 128 the 306 workgroups took 0.134 ms.)  Very long
    // bursts cap at 512 (decision strips in LDS) and take several rounds.
#ifndef VIT_CHUNK_LANES
#define VIT_CHUNK_LANES 65536    // lanes the burst is cut for: 256 workgroups x 256 lanes (A/B aid: profiles/r06_ab_viterbi_chunk_lanes.log)
#endif
    int ch = (int)((ncalls + VIT_CHUNK_LANES - 1) / VIT_CHUNK_LANES);
    ch = (ch + 15) / 16 * 16;
    if (ch < 32) ch = 32;
    if (ch > 512) ch = 512;
    const int64_t nthreads = (ncalls + ch - 1) / ch;       // chunks
    const int64_t nblocks = (nthreads + VIT_THREADS - 1) / VIT_THREADS;
    WF_REQUIRE(nblocks < (1ll << 31), "wf_viterbi4_detect: burst too long for one launch");
    hipStream_t s = wf_stream(stream);
    const size_t lds = (size_t)VIT_THREADS * (ch / 16 + 1) * sizeof(uint64_t);
    int rcv = wf_ctx_reserve_vit(ctx, vit_edge_words(nthreads));
    if (rcv) return rcv;
    double *edge = ctx->d_vit_edge;
    if (lds > 32 * 1024) {
        const void *kfn = packed ? reinterpret_cast<const void *>(viterbi_batch_kernel<true>)
                                 : reinterpret_cast<const void *>(viterbi_batch_kernel<false>);
        WF_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (packed)
        hipLaunchKernelGGL(viterbi_batch_kernel<true>, dim3((unsigned)nblocks), dim3(VIT_THREADS), lds, s, d_mf_ri, ncalls, ch,
                           differential ? 1 : 0, warmup, d_bits, d_syms, d_state, edge, nthreads);
    else
        hipLaunchKernelGGL(viterbi_batch_kernel<false>, dim3((unsigned)nblocks), dim3(VIT_THREADS), lds, s, d_mf_ri, ncalls, ch,
                           differential ? 1 : 0, warmup, d_bits, d_syms, d_state, edge, nthreads);
    WF_LAUNCH_CHECK();
    if (nthreads > 1) {
        // compare the chunk boundaries, repair what differs (viterbi_fixup_kernel); mode 0 only counts
        const unsigned fgrid = (unsigned)wf_grid_for(nthreads - 1, 256, 1024);
        const int passes = ctx->opt[WF_OPT_DET_REPAIR] == 0 && ctx->opt[WF_OPT_DET_FINAL_VERIFY] ? 2 : 1;
        for (int pass = 0; pass < passes; ++pass) {
            const int mode = pass == 0 && ctx->opt[WF_OPT_DET_REPAIR] == 0 ? 1 : 0;
            if (packed)
                hipLaunchKernelGGL(viterbi_fixup_kernel<true>, dim3(fgrid), dim3(256), 0, s, d_mf_ri, ncalls, ch, differential ? 1 : 0, d_bits, d_syms,
                                   d_state, edge, nthreads, ctx->d_vit_unmerged, mode);
            else
                hipLaunchKernelGGL(viterbi_fixup_kernel<false>, dim3(fgrid), dim3(256), 0, s, d_mf_ri, ncalls, ch, differential ? 1 : 0, d_bits, d_syms,
                                   d_state, edge, nthreads, ctx->d_vit_unmerged, mode);
            WF_LAUNCH_CHECK();
        }
    }
    if (d_state) {
        hipLaunchKernelGGL(viterbi_carry_commit_kernel, dim3(1), dim3(64), 0, s, d_state);
        WF_LAUNCH_CHECK();
    }
    return WF_OK;
}

extern "C" int wf_viterbi4_detect(wf_ctx *ctx, const double *d_mf_ri, int64_t ncalls, int differential,
                                  int warmup, uint8_t *d_bits, int8_t *d_syms, double *d_state,
                                  void *stream)
{
    return viterbi_launch(ctx, d_mf_ri, ncalls, differential, warmup, d_bits, d_syms, d_state, stream);
}

extern "C" int wf_viterbi4_unmerged(wf_ctx *ctx, int64_t *h_count, int reset, void *stream)
{
    WF_REQUIRE(ctx && h_count, "wf_viterbi4_unmerged: NULL argument");
    WF_HIP(hipSetDevice(ctx->device));
    hipStream_t s = wf_stream(stream);
    unsigned long long v = 0;
    WF_HIP(hipMemcpyAsync(&v, ctx->d_vit_unmerged, sizeof(v), hipMemcpyDeviceToHost, s));
    WF_HIP(hipStreamSynchronize(s));
    if (reset && v) WF_HIP(hipMemsetAsync(ctx->d_vit_unmerged, 0, sizeof(v), s));
    *h_count = (int64_t)v;
    return WF_OK;
}

extern "C" int wf_viterbi_repaired(wf_ctx *ctx, int64_t *h_count, int reset, void *stream)
{
    WF_REQUIRE(ctx && h_count, "wf_viterbi_repaired: NULL argument");
    WF_HIP(hipSetDevice(ctx->device));
    hipStream_t s = wf_stream(stream);
    unsigned long long v = 0;
    WF_HIP(hipMemcpyAsync(&v, ctx->d_vit_unmerged + 1, sizeof(v), hipMemcpyDeviceToHost, s));
    WF_HIP(hipStreamSynchronize(s));
    if (reset && v) WF_HIP(hipMemsetAsync(ctx->d_vit_unmerged + 1, 0, sizeof(v), s));
    *h_count = (int64_t)v;
    return WF_OK;
}

extern "C" int wf_viterbi_cascaded(wf_ctx *ctx, int64_t *h_count, int reset, void *stream)
{
    WF_REQUIRE(ctx && h_count, "wf_viterbi_cascaded: NULL argument");
    WF_HIP(hipSetDevice(ctx->device));
    hipStream_t s = wf_stream(stream);
    unsigned long long v = 0;
    WF_HIP(hipMemcpyAsync(&v, ctx->d_vit_unmerged + 2, sizeof(v), hipMemcpyDeviceToHost, s));
    WF_HIP(hipStreamSynchronize(s));
    if (reset && v) WF_HIP(hipMemsetAsync(ctx->d_vit_unmerged + 2, 0, sizeof(v), s));
    *h_count = (int64_t)v;
    return WF_OK;
}

int wf_viterbi4_detect_packed(wf_ctx *ctx, const double *d_rows4, int64_t ncalls, int differential, int warmup,
                              uint8_t *d_bits, int8_t *d_syms, double *d_state, void *stream)
{
    return viterbi_launch(ctx, d_rows4, ncalls, differential, warmup, d_bits, d_syms, d_state, stream, true);
}

extern "C" int wf_viterbi4_detect_count(wf_ctx *ctx, const double *d_mf_ri, int64_t ncalls, int differential,
                                        int warmup, uint8_t *d_bits, int8_t *d_syms, const uint8_t *d_ref_bits,
                                        const int8_t *d_ref_syms, int skip, int64_t ncompare, int64_t *d_counts,
                                        void *stream)
{
    WF_REQUIRE(d_ref_bits && d_ref_syms && d_counts && skip >= 0 && ncompare >= 0,
               "wf_viterbi4_detect_count: bad reference arguments");
    int rc = viterbi_launch(ctx, d_mf_ri, ncalls, differential, warmup, d_bits, d_syms, nullptr, stream);
    if (rc) return rc;
    int64_t m = ncalls - skip;
    if (m > ncompare) m = ncompare;
    if (m <= 0) return WF_OK;
    return wf_count_errors(ctx, d_syms + skip, d_ref_syms, d_bits + skip, d_ref_bits, m, d_counts, stream);
}

// ------------------------------------------------------------------------------------
// Batch detector for any window length 1 <= L <= VWIN_MAX_LEN (algorithm.py:19-42: `length` is a
// free parameter of the reference).  Its window loop :69-87 and depth-`length` traceback :90-98 are
// only self-consistent for even lengths: for an odd one the 8 increments of a row were computed, by list
// position, from the branch list of the OTHER trellis section than the stage that consumes them (:57-63
// uses section i % 2 at the call the row arrives, :69-87 section (i + j - 1) % 2 for the row of call
// i - L + 1 + j) — a well-defined recurrence all the same, restated literally by vwin_acs_mis below; and
// for L = 1 the single stage reads and writes the same metrics column, state by state, in place (:76-87).
// What one .iteration() call k does, restated:
//   - the window holds the rows of calls k-L+1 .. k (a row's 8 branch increments are 4 signed
//     components of it, vit_comp; rows before the burst are zeros = the zero-initialised history),
//   - entering metrics e = C - min(C), C = the stage-0 metrics the PREVIOUS call produced (:65-67),
//   - stage 0 on row k-L+1 gives the new C; stages 1 .. L-1 look ahead over the newer rows (:69-87),
//   - the decision is the stage-0 branch on the path that ends in the first arg-min state of
//     stage L-1 (:90-98, element [0] of what the call returns).
// The traceback is a register exchange here: every state carries the code (input bit, output
// symbol) of the stage-0 branch its survivor went through; an ACS stage selects the code along with
// the metric, the best end state's code is the decision — L x 4 selects instead of L dependent
// table walks.
// Parallelism as in viterbi_batch_kernel: a lane owns CH consecutive calls and re-derives C over W
// earlier calls from zero metrics (the trellis is the same 4-state one: survivors merge within a few
// tens of rows); every launch records, per chunk, the C its own calls started from and the C it
// ended with, and vwin_verify_kernel counts the chunks whose start is not BITWISE the predecessor's
// end (wf_viterbi4_unmerged) — the condition under which every decision is the sequential detector's.
// The window lives in a lane-private LDS ring of L rows (33 16-byte slots per lane at most: an odd
// stride, so the lanes' ds_read_b128 hit different banks; 128 lanes per workgroup up to L = 16, 64 above: 132 KB at
// L = 64); stage loops are rolled, so one kernel
// serves every L.  Carry block (d_state, VWIN_STATE_DOUBLES doubles, zeros = a fresh detector):
// [0] calls made, [1..4] C, [8 + 4 q ..] components of the row of call i-L+1+q, q < L-1; staging at +VWIN_STAGE.
#define VWIN_THREADS 128
#define VWIN_MAX_LEN VIT_MAX_LEN
#define VWIN_STATE_DOUBLES 1024
#define VWIN_STAGE 512
static_assert(8 + 4 * (VWIN_MAX_LEN - 1) <= VWIN_STAGE, "carry block");

template <int COL>
__device__ __forceinline__ void vwin_acs(const double m[4], const vit_comp &q, double out[4], bool lt[4])
{
    double fa[4], fb[4];
    if (COL == 0) {
        fa[0] = m[0] - q.i1; fb[0] = m[2] + q.a;
        fa[1] = m[1] - q.r1; fb[1] = m[3] + q.b;
        fa[2] = m[0] - q.b;  fb[2] = m[2] + q.r1;
        fa[3] = m[1] - q.a;  fb[3] = m[3] + q.i1;
    } else {
        fa[0] = m[0] - q.i1; fb[0] = m[1] - q.b;
        fa[1] = m[0] - q.a;  fb[1] = m[1] - q.r1;
        fa[2] = m[2] + q.r1; fb[2] = m[3] + q.a;
        fa[3] = m[2] + q.b;  fb[3] = m[3] + q.i1;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        lt[s] = fb[s] < fa[s];                    // strict '<': the first listed branch keeps a tie (algorithm.py:79-83)
        out[s] = lt[s] ? fb[s] : fa[s];
    }
}

// code of branch (start -> end e) of section col: input bit | output-symbol index << 1
__device__ __forceinline__ uint32_t vwin_code(int col, int e, int second, int diff)
{
    const int start = col == 0 ? (e & 1) + 2 * second : (e & 2) + second;
    const int b = 2 * start + (col == 0 ? e >> 1 : e & 1);
    return (uint32_t)br_inp(col, b, diff) | ((uint32_t)br_out_idx(col, b) << 1);
}

template <int COL>
__device__ __forceinline__ void vwin_stage(double m[4], uint32_t d[4], const vit_comp &q, bool first, int diff)
{
    double o[4];
    bool lt[4];
    vwin_acs<COL>(m, q, o, lt);
    uint32_t n[4];
    if (first) {
#pragma unroll
        for (int e = 0; e < 4; ++e) n[e] = lt[e] ? vwin_code(COL, e, 1, diff) : vwin_code(COL, e, 0, diff);
    } else if (COL == 0) {      // predecessors of end state e in list order: (e & 1), (e & 1) + 2
        n[0] = lt[0] ? d[2] : d[0]; n[1] = lt[1] ? d[3] : d[1]; n[2] = lt[2] ? d[2] : d[0]; n[3] = lt[3] ? d[3] : d[1];
    } else {                    // (e & 2), (e & 2) + 1
        n[0] = lt[0] ? d[1] : d[0]; n[1] = lt[1] ? d[1] : d[0]; n[2] = lt[2] ? d[3] : d[2]; n[3] = lt[3] ? d[3] : d[2];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        m[e] = o[e];
        d[e] = n[e];
    }
}

// Odd window lengths: stage section CS consumes the increments a row got, by LIST POSITION, from section 1 - CS
// (algorithm.py:74 zips `branches` of one section with a history column filled from the other, :57-63).  List
// position b of either section starts in state b >> 1; its increment Re(state_exp_term[start] * mf[idx(out)]) under
// section CR, as a signed component of the row (vit_comp, picked by the row's OWN section):
//   CR 0: -i1, -b, -r1, -a, +a, +r1, +b, +i1        CR 1: -i1, -a, -b, -r1, +r1, +b, +a, +i1
// The two positions that end in state e under CS, in list order: CS 0: b = 2 s + (e >> 1), s = (e & 1), (e & 1) + 2;
// CS 1: b = 2 s + (e & 1), s = (e & 2), (e & 2) + 1.  INPLACE (L = 1): the stage reads and writes ONE metrics
// column, state by state (:76-87 with (j - 1) % 1 == j): states 1 .. 3 see the new values of the states before them.
template <int CR>
__device__ __forceinline__ double vwin_inc(int b, const vit_comp &q)
{
    if (CR == 0) {
        switch (b) {
        case 0: return -q.i1; case 1: return -q.b; case 2: return -q.r1; case 3: return -q.a;
        case 4: return q.a; case 5: return q.r1; case 6: return q.b; default: return q.i1;
        }
    }
    switch (b) {
    case 0: return -q.i1; case 1: return -q.a; case 2: return -q.b; case 3: return -q.r1;
    case 4: return q.r1; case 5: return q.b; case 6: return q.a; default: return q.i1;
    }
}

template <int CS, bool INPLACE>
__device__ __forceinline__ void vwin_stage_mis(double m[4], uint32_t d[4], const vit_comp &q, bool first, int diff)
{
    constexpr int CR = 1 - CS;
    double o[4];
    uint32_t n[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int sa = CS == 0 ? (e & 1) : (e & 2), sb = CS == 0 ? (e & 1) + 2 : (e & 2) + 1;
        const int ba = 2 * sa + (CS == 0 ? e >> 1 : e & 1), bb = 2 * sb + (CS == 0 ? e >> 1 : e & 1);
        const double ma = INPLACE && sa < e ? o[sa] : m[sa], mb = INPLACE && sb < e ? o[sb] : m[sb];
        const double fa = ma + vwin_inc<CR>(ba, q), fb = mb + vwin_inc<CR>(bb, q);
        const bool lt = fb < fa;                  // strict '<': the first listed branch keeps a tie (algorithm.py:79-83)
        o[e] = lt ? fb : fa;
        n[e] = first ? (lt ? vwin_code(CS, e, 1, diff) : vwin_code(CS, e, 0, diff)) : (lt ? d[sb] : d[sa]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        m[e] = o[e];
        d[e] = n[e];
    }
}

// One chunk of the window detector by one thread (ring: this thread's (2 L + 1) 16-byte slots of LDS).  start == nullptr:
// the first launch — W warm-up calls from zero metrics (or, for the chunk at the head of the burst, the carried
// detector).  start != nullptr: a repair (vwin_fixup_kernel) — the chunk's own calls again from those C, decisions and
// end record rewritten; returns true when the chunk's END changed.
__device__ __forceinline__ bool vwin_chunk(const double2 *__restrict__ rows, int64_t ncalls, int L, int CH, int W, int diff,
                                           uint8_t *__restrict__ bits, int8_t *__restrict__ syms, double *__restrict__ state,
                                           double *__restrict__ erec, double2 *__restrict__ ring, int64_t chunk, const double *start)
{
    const int64_t a = chunk * CH;
    const bool live = a < ncalls;
    const int64_t i0 = state ? (int64_t)state[0] : 0;                 // calls made before this launch
    const int64_t kend = live ? (a + CH < ncalls ? a + CH : ncalls) : 0;
    const bool exact = start != nullptr || a <= W;                    // starts from the true C: a repair, or call 0 from the carried detector itself
    const int64_t ks = start ? a : (exact ? 0 : a - W);               // even: CH and W are even
    double C[4] = {0.0, 0.0, 0.0, 0.0};
    if (start) {
#pragma unroll
        for (int q = 0; q < 4; ++q) C[q] = start[q];
    } else if (exact && state && i0 > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) C[q] = state[1 + q];
    }
    // components of the row of local call kp (its trellis section = parity of its global call index)
    auto comps = [&](int64_t kp) __attribute__((always_inline)) {
        vit_comp c;
        if (kp >= 0) {
            const int64_t kk = kp < ncalls ? kp : ncalls - 1;         // (clamped rows are never decoded)
            const double2 z0 = vit_ld16(rows + 3 * kk), z1 = vit_ld16(rows + 3 * kk + 1), z2 = vit_ld16(rows + 3 * kk + 2);
            const bool odd = ((i0 + kp) & 1) != 0;
            c.r1 = z1.x; c.i1 = z1.y;
            c.a = odd ? z0.y : z0.x;
            c.b = odd ? z2.x : z2.y;
        } else if (state && i0 > 0) {                                 // before this launch: the carried window
            const double *p = state + 8 + 4 * (L - 1 + kp);
            c.r1 = p[0]; c.i1 = p[1]; c.a = p[2]; c.b = p[3];
        } else {
            c.r1 = c.i1 = c.a = c.b = 0.0;                            // the reference's zero-initialised history
        }
        return c;
    };
    auto slot = [&](int64_t kp) __attribute__((always_inline)) { return (int)(((kp % L) + L) % L); };
    auto put = [&](int64_t kp, const vit_comp &c) __attribute__((always_inline)) {
        const int q = slot(kp);
        ring[2 * q] = make_double2(c.r1, c.i1);
        ring[2 * q + 1] = make_double2(c.a, c.b);
    };
    auto get = [&](int q) __attribute__((always_inline)) {
        const double2 u = ring[2 * q], v = ring[2 * q + 1];
        vit_comp c;
        c.r1 = u.x; c.i1 = u.y; c.a = v.x; c.b = v.y;
        return c;
    };
    if (live)
        for (int q = 1; q < L; ++q) put(ks - L + q, comps(ks - L + q));
    vit_comp nxt = comps(ks);
    const int nsteps = start ? CH : W + CH;                           // common trip count; a lane is active while k < kend
    const int par = (int)((i0 + 1) & 1);                              // section of stage 0 at step t: (i0 + ks + t - 1) & 1 (algorithm.py:71 at j = 0), ks even
    const bool mis = (L & 1) != 0;                                    // odd L: a row's increments come from the other section than its stage
    for (int t = 0; t < nsteps; ++t) {
        const int64_t k = ks + t;
        const bool act = live && k < kend;                            // (no barrier below: inactive lanes just idle)
        if (!act) continue;
        put(k, nxt);
        nxt = comps(k + 1);                                           // next call's row: in flight during this call's stages
        if (k == a && erec) {
#pragma unroll
            for (int q = 0; q < 4; ++q) erec[8 * chunk + q] = C[q];
        }
        const double mn = fmin(fmin(C[0], C[1]), fmin(C[2], C[3]));
        double m[4] = {C[0] - mn, C[1] - mn, C[2] - mn, C[3] - mn};
        uint32_t d[4] = {0u, 0u, 0u, 0u};
        int q = slot(k - L + 1);
        const bool odd0 = ((par + t) & 1) != 0;                       // wave-uniform
        // stages in (stage 0's section, the other one) pairs, and for an odd L one more of stage 0's section
        for (int j = 0; j + 1 < L; j += 2) {
            const vit_comp r0 = get(q);
            q = q + 1 == L ? 0 : q + 1;
            const vit_comp r1 = get(q);
            q = q + 1 == L ? 0 : q + 1;
            if (!mis) {
                if (!odd0) {
                    vwin_stage<0>(m, d, r0, j == 0, diff);
                    if (j == 0) { C[0] = m[0]; C[1] = m[1]; C[2] = m[2]; C[3] = m[3]; }
                    vwin_stage<1>(m, d, r1, false, diff);
                } else {
                    vwin_stage<1>(m, d, r0, j == 0, diff);
                    if (j == 0) { C[0] = m[0]; C[1] = m[1]; C[2] = m[2]; C[3] = m[3]; }
                    vwin_stage<0>(m, d, r1, false, diff);
                }
            } else if (!odd0) {
                vwin_stage_mis<0, false>(m, d, r0, j == 0, diff);
                if (j == 0) { C[0] = m[0]; C[1] = m[1]; C[2] = m[2]; C[3] = m[3]; }
                vwin_stage_mis<1, false>(m, d, r1, false, diff);
            } else {
                vwin_stage_mis<1, false>(m, d, r0, j == 0, diff);
                if (j == 0) { C[0] = m[0]; C[1] = m[1]; C[2] = m[2]; C[3] = m[3]; }
                vwin_stage_mis<0, false>(m, d, r1, false, diff);
            }
        }
        if (mis) {
            const vit_comp r0 = get(q);
            if (L == 1) {
                if (!odd0) vwin_stage_mis<0, true>(m, d, r0, true, diff);
                else vwin_stage_mis<1, true>(m, d, r0, true, diff);
                C[0] = m[0]; C[1] = m[1]; C[2] = m[2]; C[3] = m[3];
            } else if (!odd0) {
                vwin_stage_mis<0, false>(m, d, r0, false, diff);
            } else {
                vwin_stage_mis<1, false>(m, d, r0, false, diff);
            }
        }
        if (k >= a) {
            // np.argmin: the first minimum
            double best = m[0];
            uint32_t dec = d[0];
#pragma unroll
            for (int e = 1; e < 4; ++e) {
                const bool lt = m[e] < best;
                best = lt ? m[e] : best;
                dec = lt ? d[e] : dec;
            }
            bits[k] = (uint8_t)(dec & 1u);
            syms[k] = (int8_t)(2 * (int)((dec >> 1) & 3u) - 2);
        }
    }
    bool changed = false;
    if (live && erec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (start) changed |= __double_as_longlong(erec[8 * chunk + 4 + q]) != __double_as_longlong(C[q]);
            erec[8 * chunk + 4 + q] = C[q];
        }
    }
    if (live && state && kend == ncalls && (!start || changed)) {     // the lane that owns the last call: carry out (staging)
        state[VWIN_STAGE] = (double)(i0 + ncalls);
#pragma unroll
        for (int q = 0; q < 4; ++q) state[VWIN_STAGE + 1 + q] = C[q];
        for (int q = 0; q < L - 1; ++q) {
            const int64_t kp = ncalls - (L - 1) + q;
            // rows of this launch are in the ring when this lane walked them; older ones come from the inputs again
            const vit_comp c = (kp >= ks - L + 1) ? get(slot(kp)) : comps(kp);
            double *p = state + VWIN_STAGE + 8 + 4 * q;
            p[0] = c.r1; p[1] = c.i1; p[2] = c.a; p[3] = c.b;
        }
    }
    return changed;
}

__global__ __launch_bounds__(VWIN_THREADS) void viterbi_window_kernel(const double2 *__restrict__ rows, int64_t ncalls, int L, int CH, int W,
                                                                       int diff, uint8_t *__restrict__ bits, int8_t *__restrict__ syms,
                                                                       double *__restrict__ state, double *__restrict__ erec, int64_t nchunks)
{
    extern __shared__ __attribute__((aligned(16))) double2 s_ring[];
    if (blockIdx.x == 0 && threadIdx.x < VIT_HDR) reinterpret_cast<uint64_t *>(erec + 8 * nchunks)[threadIdx.x] = 0;   // lists empty, nobody arrived
    const int64_t chunk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    (void)vwin_chunk(rows, ncalls, L, CH, W, diff, bits, syms, state, erec, s_ring + (size_t)threadIdx.x * (2 * L + 1), chunk, nullptr);
}

// The fix-up launch of the window detector (layout and rounds: see viterbi_fixup_kernel's comment).
__global__ __launch_bounds__(VWIN_THREADS) void vwin_fixup_kernel(const double2 *__restrict__ rows, int64_t ncalls, int L, int CH, int W, int diff,
                                                                   uint8_t *__restrict__ bits, int8_t *__restrict__ syms, double *__restrict__ state,
                                                                   double *__restrict__ erec, int64_t nchunks, unsigned long long *__restrict__ unmerged,
                                                                   int mode)
{
    extern __shared__ __attribute__((aligned(16))) double2 s_ring[];
    if (!vit_fixup_verify(erec, nchunks, unmerged, mode)) return;
    double2 *ring = s_ring + (size_t)threadIdx.x * (2 * L + 1);
    vit_fixup_rounds(erec, nchunks, unmerged, [&](int64_t c) {
        double st[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            st[q] = __hip_atomic_load(erec + 8 * (c - 1) + 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the predecessor's end, as it is now
            erec[8 * c + q] = st[q];                       // ... is what this run of the chunk starts from
        }
        return vwin_chunk(rows, ncalls, L, CH, W, diff, bits, syms, state, erec, ring, c, st);
    });
}

__global__ void vwin_carry_commit_kernel(double *state, int n)
{
    for (int t = threadIdx.x; t < n; t += blockDim.x) state[t] = state[VWIN_STAGE + t];
}

extern "C" int wf_viterbi4_detect_window(wf_ctx *ctx, const double *d_mf_ri, int64_t ncalls, int length, int differential,
                                         int warmup, uint8_t *d_bits, int8_t *d_syms, double *d_state, void *stream)
{
    WF_REQUIRE(ctx && ncalls >= 0 && warmup >= 0, "wf_viterbi4_detect_window: bad argument");
    WF_REQUIRE(length >= 1 && length <= VWIN_MAX_LEN, "wf_viterbi4_detect_window: window length %d (1 .. %d)", length, VWIN_MAX_LEN);
    if (ncalls == 0) return WF_OK;
    WF_REQUIRE(d_mf_ri && d_bits && d_syms, "wf_viterbi4_detect_window: NULL device pointer");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_mf_ri) & 15) == 0, "wf_viterbi4_detect_window: rows must be 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    // warm-up: the merge depth of the 4-state trellis (the L = 2 kernel's scan: 16 .. 48 rows never left a chunk
    // unproven) — the look-ahead does not change which C a chunk converges to, only how late its decisions come
    int W = warmup ? warmup : VIT_DEFAULT_WARMUP + 1;
    W = (W + 1) / 2 * 2;
    if (W > 4096) W = 4096;
    // calls per lane: even, enough lanes for ~2 waves per SIMD on a long burst, the warm-up a fraction of the work
    int64_t ch = (ncalls + 131071) / 131072;
    ch = (ch + 1) / 2 * 2;
    if (ch < 4 * W) ch = 4 * W;
    if (ch > (1 << 20)) ch = 1 << 20;
    const int64_t nchunks = (ncalls + ch - 1) / ch;
    const int threads = length <= 16 ? VWIN_THREADS : 64;               // lanes per workgroup: the window ring is (2 L + 1) x 16 B of LDS per lane
    const int64_t nblocks = (nchunks + threads - 1) / threads;
    WF_REQUIRE(nblocks < (1ll << 31), "wf_viterbi4_detect_window: burst too long for one launch");
    int rc = wf_ctx_reserve_vit(ctx, vit_edge_words(nchunks));
    if (rc) return rc;
    hipStream_t s = wf_stream(stream);
    const size_t lds = (size_t)threads * (2 * length + 1) * sizeof(double2);
    if (lds > 48 * 1024) {
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(viterbi_window_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(vwin_fixup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(viterbi_window_kernel, dim3((unsigned)nblocks), dim3(threads), lds, s, reinterpret_cast<const double2 *>(d_mf_ri),
                       ncalls, length, (int)ch, W, differential ? 1 : 0, d_bits, d_syms, d_state, ctx->d_vit_edge, nchunks);
    WF_LAUNCH_CHECK();
    if (nchunks > 1) {
        // compare the chunk boundaries and repair what differs (vwin_fixup_kernel); mode 0 only counts
        const unsigned fgrid = (unsigned)wf_grid_for(nchunks - 1, threads, 1024);
        const int passes = ctx->opt[WF_OPT_DET_REPAIR] == 0 && ctx->opt[WF_OPT_DET_FINAL_VERIFY] ? 2 : 1;
        for (int pass = 0; pass < passes; ++pass) {
            hipLaunchKernelGGL(vwin_fixup_kernel, dim3(fgrid), dim3(threads), lds, s, reinterpret_cast<const double2 *>(d_mf_ri), ncalls, length, (int)ch,
                               W, differential ? 1 : 0, d_bits, d_syms, d_state, ctx->d_vit_edge, nchunks, ctx->d_vit_unmerged,
                               pass == 0 && ctx->opt[WF_OPT_DET_REPAIR] == 0 ? 1 : 0);
            WF_LAUNCH_CHECK();
        }
    }
    if (d_state) {
        hipLaunchKernelGGL(vwin_carry_commit_kernel, dim3(1), dim3(128), 0, s, d_state, 8 + 4 * (length - 1));
        WF_LAUNCH_CHECK();
    }
    return WF_OK;
}

extern "C" int64_t wf_viterbi4_window_state_bytes(void) { return (int64_t)(VWIN_STATE_DOUBLES * sizeof(double)); }

// ------------------------------------------------------------------------------------
// Literal single-call form for any window length (the per-symbol drop-in API).
struct vit_state {
    long long i;
    double bi_history[8][VIT_MAX_LEN];
    double metrics[4][VIT_MAX_LEN];
    unsigned char path[4][VIT_MAX_LEN];
};

extern "C" int64_t wf_viterbi4_state_bytes(int length)
{
    return (length >= 1 && length <= VIT_MAX_LEN) ? (int64_t)sizeof(vit_state) : -1;
}

// One literal .iteration() (algorithm.py:44-101) by ONE thread.  mf3 / bits_out / syms_out may be host memory the
// device addresses directly (volatile: the mailbox form below re-reads them on every request).
__device__ __forceinline__ void vit_iteration_body(vit_state *st, int L, int diff, const volatile double *mf3,
                                                   volatile double *bits_out, volatile double *syms_out)
{
    const long long i = st->i;
    const int col_now = (int)(((i % 2) + 2) % 2);
    // algorithm.py:57-63: np.roll(-1) then overwrite the last column
    for (int b = 0; b < 8; ++b) {
        for (int j = 0; j + 1 < L; ++j) st->bi_history[b][j] = st->bi_history[b][j + 1];
        const int oi = br_out_idx(col_now, b);
        st->bi_history[b][L - 1] = br_inc(b >> 1, mf3[2 * oi], mf3[2 * oi + 1]);
    }
    // algorithm.py:65-67
    double mn = st->metrics[0][0];
    for (int s = 1; s < 4; ++s) mn = st->metrics[s][0] < mn ? st->metrics[s][0] : mn;
    double carried[4];
    for (int s = 0; s < 4; ++s) carried[s] = st->metrics[s][0] - mn;
    for (int s = 0; s < 4; ++s) {
        for (int j = 0; j + 1 < L; ++j) st->metrics[s][j] = 0.0;
        st->metrics[s][L - 1] = carried[s];
        for (int j = 0; j < L; ++j) st->path[s][j] = 0;
    }
    // algorithm.py:69-87 (in-place, stage by stage, end state by end state)
    for (int j = 0; j < L; ++j) {
        const int col = (int)((((i + j - 1) % 2) + 2) % 2);
        const int jm1 = (j - 1 + L) % L;
        for (int s = 0; s < 4; ++s) {
            int min_k = 0;
            double min_m = INFINITY;
            for (int b = 0; b < 8; ++b) {
                if (br_end(col, b) != s) continue;
                const double mm = st->metrics[b >> 1][jm1] + st->bi_history[b][j];
                if (mm < min_m) {
                    min_m = mm;
                    min_k = b >> 1;
                }
            }
            st->metrics[s][j] = min_m;
            st->path[s][j] = (unsigned char)min_k;
        }
    }
    // algorithm.py:90-98
    int state = 0;
    for (int s = 1; s < 4; ++s)
        if (st->metrics[s][L - 1] < st->metrics[state][L - 1]) state = s;
    for (int j = L - 1; j >= 0; --j) {
        const int col = (int)((((i + j - 1) % 2) + 2) % 2);
        const int pred = st->path[state][j];
        const int b = 2 * pred + (col == 0 ? (state >> 1) : (state & 1));
        // reverse_transitions lookup: the branch must really end in `state`
        if (br_end(col, b) != state) {
            bits_out[j] = nan("");
            syms_out[j] = nan("");
        } else {
            bits_out[j] = (double)br_inp(col, b, diff);
            syms_out[j] = (double)(2 * (int)br_out_idx(col, b) - 2);
        }
        state = pred;
    }
    st->i = i + 1;
}

__global__ void viterbi_iteration_kernel(vit_state *st, int L, int diff, const double *__restrict__ mf3,
                                         double *__restrict__ bits_out, double *__restrict__ syms_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    vit_iteration_body(st, L, diff, mf3, bits_out, syms_out);
}

__device__ __forceinline__ void vit_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same call by one WAVE on a detector state in LDS (the persistent server below): the loops over the 8
// branches and the 4 end states run across lanes — same operations on the same operands, so the same bits — and
// only the traceback (a dependent walk) stays with lane 0.  L >= 2 (for L = 1 the reference's in-place stage
// update reads metrics it has just written: the serial form above keeps that literal).  All 64 lanes call it.
__device__ __forceinline__ void vit_iteration_wave(vit_state *st, int L, int diff, const double (&mf3)[6], double *bits_out, double *syms_out)
{
    const int lane = threadIdx.x & 63;
    const long long i = st->i;
    const int col_now = (int)(((i % 2) + 2) % 2);
    double mn = 0.0, mine = 0.0;
    if (lane < 4) {                                    // algorithm.py:65-67: read before anything is overwritten
        mn = st->metrics[0][0];
        for (int s = 1; s < 4; ++s) mn = st->metrics[s][0] < mn ? st->metrics[s][0] : mn;
        mine = st->metrics[lane][0] - mn;
    }
    if (lane < 8) {                                    // algorithm.py:57-63, branch b = lane
        const int b = lane;
        for (int j = 0; j + 1 < L; ++j) st->bi_history[b][j] = st->bi_history[b][j + 1];
        const int oi = br_out_idx(col_now, b);
        st->bi_history[b][L - 1] = br_inc(b >> 1, mf3[2 * oi], mf3[2 * oi + 1]);
    }
    if (lane < 4) {
        for (int j = 0; j + 1 < L; ++j) st->metrics[lane][j] = 0.0;
        st->metrics[lane][L - 1] = mine;
        for (int j = 0; j < L; ++j) st->path[lane][j] = 0;
    }
    vit_wave_sync();
    for (int j = 0; j < L; ++j) {                      // algorithm.py:69-87, end state s = lane
        const int col = (int)((((i + j - 1) % 2) + 2) % 2);
        const int jm1 = (j - 1 + L) % L;               // != j for L >= 2: a stage reads the previous stage's column only
        if (lane < 4) {
            // the two branches that end in state s, in list order (the serial form's `for b ... if (br_end(col, b) != s)
            // continue` visits exactly these, ascending): column 0: start & 1 = s & 1, b & 1 = s >> 1;
            // column 1: start & 2 = s & 2, b & 1 = s & 1.  Both operand pairs are fetched before the first compare.
            const int s = lane;
            const int b0 = col == 0 ? 2 * (s & 1) + (s >> 1) : 2 * (s & 2) + (s & 1);
            const int b1 = b0 + (col == 0 ? 4 : 2);
            const double mm0 = st->metrics[b0 >> 1][jm1] + st->bi_history[b0][j];
            const double mm1 = st->metrics[b1 >> 1][jm1] + st->bi_history[b1][j];
            int min_k = 0;
            double min_m = INFINITY;
            if (mm0 < min_m) {
                min_m = mm0;
                min_k = b0 >> 1;
            }
            if (mm1 < min_m) {
                min_m = mm1;
                min_k = b1 >> 1;
            }
            st->metrics[s][j] = min_m;
            st->path[s][j] = (unsigned char)min_k;
        }
        vit_wave_sync();
    }
    if (lane == 0) {                                   // algorithm.py:90-98
        int state = 0;
        {   // (the four final metrics in one batch of reads, then np.argmin's first-minimum rule on registers)
            const double f0 = st->metrics[0][L - 1], f1 = st->metrics[1][L - 1], f2 = st->metrics[2][L - 1], f3 = st->metrics[3][L - 1];
            double best = f0;
            if (f1 < best) { best = f1; state = 1; }
            if (f2 < best) { best = f2; state = 2; }
            if (f3 < best) { best = f3; state = 3; }
        }
        for (int j = L - 1; j >= 0; --j) {
            const int col = (int)((((i + j - 1) % 2) + 2) % 2);
            const int pred = st->path[state][j];
            const int b = 2 * pred + (col == 0 ? (state >> 1) : (state & 1));
            if (br_end(col, b) != state) {
                bits_out[j] = nan("");
                syms_out[j] = nan("");
            } else {
                bits_out[j] = (double)br_inp(col, b, diff);
                syms_out[j] = (double)(2 * (int)br_out_idx(col, b) - 2);
            }
            state = pred;
        }
        st->i = i + 1;
    }
    vit_wave_sync();
}

// ---- the per-symbol call without a launch per symbol ---------------------------------------------------
// examples/soqpsk_detection.py:189-198 calls the detector once per symbol from a Python loop.  A kernel
// launch + a stream synchronise per call cost 22.7 us (profiles/r02_iteration_bench.json), 2.5x fewer than the
// reference's 57 us of interpreted Python but all of it launch latency.  Instead ONE single-wave kernel is
// started by the first call and serves every following call through a mailbox in pinned host memory that the
// device addresses directly: the host writes the three matched-filter outputs and bumps `req`, the kernel (lane
// 0 polls with system-scope atomics) runs the iteration on the detector state in device memory, writes the
// 2 x length results and sets `ack` = `req`; the host spins on `ack`.  Two PCIe round trips per call, no launch.
// EXIT CONDITIONS (every wave reaches one): `stop` set by the host (context teardown), or no request for
// VIT_SERVER_IDLE_TICKS of the constant-rate wall clock (10 ms): the kernel then clears `running` and retires;
// the next call simply starts it again (the detector state lives in device memory, the sequence numbers in the
// mailbox).  It runs on the context's own stream, so nothing the caller queues waits behind it, and a
// device-wide synchronise waits for at most the idle time.
struct vit_mailbox {
    unsigned long long req, ack, stop, running;      // sequence numbers / flags (each written by ONE side)
    // The request: eight 16-byte chunks {8 bytes of payload, the request's sequence number}, each written by the
    // host with ONE 16-byte store and read by the device in ONE burst of eight loads — the poll IS the read: a
    // burst whose eight tags agree on a new number is a complete request (a reader of host memory pays ~4.5 us per
    // round trip here, so a separate "has something arrived" word would double the latency).
    // payload: [0] vit_state* of the detector, [1] length | diff << 32, [2..7] the three matched-filter outputs
    unsigned long long pad_[12];                     // (the chunks start a 128-byte line: ONE wave-wide load fetches them)
    unsigned long long chunk[8][2];
    // The answer, tagged the same way: chunk k < length = {bits[k], sequence number}, chunk length + k = {syms[k], ...},
    // each ONE 16-byte store of the device.  The host takes the answer when all 2 x length tags carry its request's
    // number — no fence and no acknowledgement word on the way back; the detector state is written through to device
    // memory AFTER the answer has left (the acknowledgement below only tells the host that this, too, is done).
    unsigned long long ans[2 * VIT_MAX_LEN][2];
    unsigned long long t_seen, t_read, t_state, t_body, t_done;   // wall_clock64() ticks (100 MHz) of the last request (tools/iteration_bench.py)
};
static_assert(offsetof(vit_mailbox, chunk) == 128, "request chunks must start a 128-byte line");
#define VIT_SERVER_IDLE_TICKS 1000000ull             // wall_clock64() ticks at 100 MHz: 10 ms

__global__ void viterbi_iteration_server_kernel(vit_mailbox *mb)
{
    // The detector state the requests are for is CACHED in LDS between requests (6.4 KB; a single thread walking it
    // in device memory paid a ~1 us dependent load per access: 20 .. 30 us per call).  Write-through: what a call
    // changed (the first `length` columns of the three tables and the counter) goes back to device memory before
    // the acknowledgement, so the home copy is always current — nothing is written when the server retires, and a
    // detector whose memory has been freed and handed to a NEW one is never clobbered.  The cache is valid while
    // the home's call counter equals the cached one (a new, zero-filled detector at the same address does not).
    __shared__ vit_state s_st;
    __shared__ double s_out[2 * VIT_MAX_LEN];
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    vit_state *home = nullptr;
    unsigned long long last = __hip_atomic_load(&mb->ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long idle_since = wall_clock64();
    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
    const volatile v2u *rq = reinterpret_cast<const volatile v2u *>(&mb->chunk[0][0]);
    for (;;) {     // every lane runs the same loop on the same (uniform) values; lane 0 alone writes to the mailbox
        const unsigned long long t_poll = wall_clock64();
        // ONE load instruction: lanes 0 .. 7 fetch one 16-byte chunk each = one contiguous 128-byte line (a load of
        // host memory costs ~1.1 us EACH when issued one after another; 64 lanes issuing eight loads apiece cost 9 us);
        // the chunks reach every lane as scalars
        v2u mine = {0, 0};
        if (lane < 8) mine = rq[lane];
        auto pick = [&](int k) __attribute__((always_inline)) {
            v2u r;
            r.x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine.x >> 32), k) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mine.x, k);
            r.y = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(mine.y >> 32), k) << 32) | (unsigned)__builtin_amdgcn_readlane((int)mine.y, k);
            return r;
        };
        const v2u c0 = pick(0), c1 = pick(1), c2 = pick(2), c3 = pick(3), c4 = pick(4), c5 = pick(5), c6 = pick(6), c7 = pick(7);
        const unsigned long long seq = c0.y;
        const bool whole = c1.y == seq && c2.y == seq && c3.y == seq && c4.y == seq && c5.y == seq && c6.y == seq && c7.y == seq;
        if (seq != last && whole) {
            const unsigned long long t_read = wall_clock64();
            const int L = (int)(c1.x & 0xFFFFFFFFull), diff = (int)(c1.x >> 32);
            const double mf[6] = {__longlong_as_double((long long)c2.x), __longlong_as_double((long long)c3.x),
                                  __longlong_as_double((long long)c4.x), __longlong_as_double((long long)c5.x),
                                  __longlong_as_double((long long)c6.x), __longlong_as_double((long long)c7.x)};
            vit_state *want = reinterpret_cast<vit_state *>(c0.x);
            if (want != home || *(volatile long long *)&want->i != *(volatile long long *)&s_st.i) {
                vit_wave_sync();
                const unsigned long long *src = reinterpret_cast<const unsigned long long *>(want);
                unsigned long long *dst = reinterpret_cast<unsigned long long *>(&s_st);
                for (int k = lane; k < (int)(sizeof(vit_state) / 8); k += 64) dst[k] = src[k];
                home = want;
                vit_wave_sync();
            }
            const unsigned long long t_state = wall_clock64();
            if (L >= 2) {
                vit_iteration_wave(&s_st, L, diff, mf, s_out, s_out + VIT_MAX_LEN);
            } else {
                if (lane == 0) vit_iteration_body(&s_st, L, diff, mf, s_out, s_out + VIT_MAX_LEN);
                vit_wave_sync();
            }
            const unsigned long long t_body = wall_clock64();
            // the answer first: 2 L tagged 16-byte chunks (lanes 0 .. 2L-1, and again from lane 64 on for L > 32) ...
            {
                volatile v2u *an = reinterpret_cast<volatile v2u *>(&mb->ans[0][0]);
                for (int k = lane; k < 2 * L; k += 64) {
                    const double v = k < L ? s_out[k] : s_out[VIT_MAX_LEN + (k - L)];
                    v2u c;
                    c.x = (unsigned long long)__double_as_longlong(v);
                    c.y = seq;
                    an[k] = c;
                }
            }
            const unsigned long long t_ans = wall_clock64();
            // ... then the write-through (posted stores), lanes over the 16 table rows, and the acknowledgement behind it
            if (lane < 8) for (int j = 0; j < L; ++j) home->bi_history[lane][j] = s_st.bi_history[lane][j];
            else if (lane < 12) for (int j = 0; j < L; ++j) home->metrics[lane - 8][j] = s_st.metrics[lane - 8][j];
            else if (lane < 16) for (int j = 0; j < L; ++j) home->path[lane - 12][j] = s_st.path[lane - 12][j];
            if (lane == 0) home->i = s_st.i;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");               // (system scope) every lane's stores before lane 0's acknowledgement
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                mb->t_seen = t_poll; mb->t_read = t_read; mb->t_state = t_state; mb->t_body = t_body; mb->t_done = t_ans;
                __hip_atomic_store(&mb->ack, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            last = seq;
            idle_since = wall_clock64();
            continue;
        }
        if (__hip_atomic_load(&mb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        if (wall_clock64() - idle_since > VIT_SERVER_IDLE_TICKS) break;
    }
    if (lane == 0) __hip_atomic_store(&mb->running, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Timing of the last served request in microseconds: {request read, state check, body, write-through + answer}
// (diagnostic for tools/iteration_bench.py; zeros when no server has run).
extern "C" int wf_viterbi4_iteration_server_timing(wf_ctx *ctx, double *h_us4)
{
    WF_REQUIRE(ctx && h_us4, "wf_viterbi4_iteration_server_timing: NULL argument");
    for (int k = 0; k < 4; ++k) h_us4[k] = 0.0;
    if (!ctx->h_mailbox) return WF_OK;
    const vit_mailbox *mb = static_cast<const vit_mailbox *>(ctx->h_mailbox);
    h_us4[0] = (double)(mb->t_read - mb->t_seen) * 0.01;      // (the poll that found the request: one burst of eight loads)
    h_us4[1] = (double)(mb->t_state - mb->t_read) * 0.01;
    h_us4[2] = (double)(mb->t_body - mb->t_state) * 0.01;
    h_us4[3] = (double)(mb->t_done - mb->t_body) * 0.01;
    return WF_OK;
}

// One request at a time, and nothing toggles the mailbox's control words under a request: the per-symbol entry point,
// the quiesce and the stop all take the context's iteration lock (Python threads drop the interpreter lock inside a
// ctypes call; a daemon thread may still be inside iteration() when the interpreter exits).
struct vit_turn {
    std::atomic_flag &f;
    explicit vit_turn(std::atomic_flag &f_) : f(f_) { while (f.test_and_set(std::memory_order_acquire)) __builtin_ia32_pause(); }
    ~vit_turn() { f.clear(std::memory_order_release); }
};

static inline double vit_now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
#define VIT_HOST_SPIN_SECONDS 5.0   // how long the host waits for the server before it reports the device gone

// Stop the server of a context (harmless when none is running): it retires within one poll.  closing: the context
// is being retired (interpreter exit / destroy) — wf_viterbi4_iteration_host will not start a new server on it.
static int vit_server_stop_locked(wf_ctx *ctx, bool closing)
{
    if (closing) ctx->iter_closing = true;
    if (!ctx->h_mailbox) return WF_OK;
    vit_mailbox *mb = static_cast<vit_mailbox *>(ctx->h_mailbox);
    __atomic_store_n(&mb->stop, 1ull, __ATOMIC_RELEASE);
    if (ctx->iter_stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(ctx->iter_stream));
    __atomic_store_n(&mb->stop, 0ull, __ATOMIC_RELEASE);
    return WF_OK;
}

int wf_iter_server_stop(wf_ctx *ctx)
{
    if (!ctx) return WF_OK;
    vit_turn my_turn(ctx->iter_lock);
    return vit_server_stop_locked(ctx, true);
}

extern "C" int wf_viterbi4_iteration_quiesce(wf_ctx *ctx);
extern "C" int wf_viterbi4_iteration(wf_ctx *ctx, void *d_state, int length, int differential,
                                     const double *d_mf3_ri, double *d_bits_out, double *d_syms_out,
                                     void *stream)
{
    WF_REQUIRE(ctx && d_state && d_mf3_ri && d_bits_out && d_syms_out, "wf_viterbi4_iteration: NULL argument");
    WF_REQUIRE(length >= 1 && length <= VIT_MAX_LEN, "wf_viterbi4_iteration: length %d", length);
    WF_HIP(hipSetDevice(ctx->device));
    {   // the state may be one the per-symbol server was asked about: its write-through must be in first
        const int rq = wf_viterbi4_iteration_quiesce(ctx);
        if (rq) return rq;
    }
    hipLaunchKernelGGL(viterbi_iteration_kernel, dim3(1), dim3(64), 0, wf_stream(stream),
                       static_cast<vit_state *>(d_state), length, differential ? 1 : 0, d_mf3_ri,
                       d_bits_out, d_syms_out);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// The detector's public arrays (algorithm.py:25-42: every instance carries bi_history f64[8][length], metrics f64[4][length],
// path u8[4][length]; a caller may look at them between iteration() calls): read from the device-resident state into host
// arrays of exactly those shapes, after the per-symbol server's write-through has landed.  Synchronous.
extern "C" int wf_viterbi4_state_read(wf_ctx *ctx, const void *d_state, int length, int64_t *h_calls, double *h_bi_history,
                                      double *h_metrics, uint8_t *h_path, void *stream)
{
    WF_REQUIRE(ctx && d_state && h_bi_history && h_metrics && h_path, "wf_viterbi4_state_read: NULL argument");
    WF_REQUIRE(length >= 1 && length <= VIT_MAX_LEN, "wf_viterbi4_state_read: length %d", length);
    WF_HIP(hipSetDevice(ctx->device));
    const int rq = wf_viterbi4_iteration_quiesce(ctx);
    if (rq) return rq;
    static_assert(sizeof(vit_state) % 8 == 0, "state block is whole words");
    vit_state *h = new vit_state;
    const hipError_t e1 = hipMemcpyAsync(h, d_state, sizeof(vit_state), hipMemcpyDeviceToHost, wf_stream(stream));
    const hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(wf_stream(stream)) : e1;
    if (e2 != hipSuccess) {
        delete h;
        WF_HIP(e2);
    }
    if (h_calls) *h_calls = (int64_t)h->i;
    for (int b = 0; b < 8; ++b)
        for (int j = 0; j < length; ++j) h_bi_history[b * length + j] = h->bi_history[b][j];
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < length; ++j) {
            h_metrics[s * length + j] = h->metrics[s][j];
            h_path[s * length + j] = h->path[s][j];
        }
    delete h;
    return WF_OK;
}

// The per-symbol call as ONE host call with host operands (the drop-in loop of
// examples/soqpsk_detection.py:189-198 calls the detector once per symbol): the three
// matched-filter outputs and the 2 x length results travel through pinned host memory that the
// device addresses directly, so a call is one kernel launch and one stream synchronise — no
// device allocation, no separate H2D / D2H copies.  Synchronous.
// one aligned 16-byte store (host): {lo, hi} — an aligned SSE store is a single 16-byte write on every x86-64 part
static inline void wf_store16(unsigned long long *p, unsigned long long lo, unsigned long long hi)
{
    typedef long long v2ll __attribute__((vector_size(16)));
    const v2ll v = {(long long)lo, (long long)hi};
    *reinterpret_cast<volatile v2ll *>(p) = v;
}

// Wait until the server has written the last served call's state through to device memory (it answers first):
// before anything else reads or frees a detector state the server may have been asked about.  Harmless without a server.
static int vit_quiesce_locked(wf_ctx *ctx)
{
    if (!ctx->h_mailbox) return WF_OK;
    vit_mailbox *mb = static_cast<vit_mailbox *>(ctx->h_mailbox);
    const double t0 = vit_now_s();
    for (unsigned spins = 0; __atomic_load_n(&mb->ack, __ATOMIC_ACQUIRE) != mb->req; ++spins) {
        if (__atomic_load_n(&mb->running, __ATOMIC_ACQUIRE) == 0 || ((spins & 0xFFFFu) == 0xFFFFu && vit_now_s() - t0 > VIT_HOST_SPIN_SECONDS)) {
            // (a retired server has nothing in flight: it acknowledges before it leaves)
            if (__atomic_load_n(&mb->ack, __ATOMIC_ACQUIRE) == mb->req) break;
            wf_set_error("wf_viterbi4_iteration_quiesce: the iteration server did not acknowledge its last call");
            return WF_ERR_HIP;
        }
        __builtin_ia32_pause();
    }
    return WF_OK;
}

extern "C" int wf_viterbi4_iteration_quiesce(wf_ctx *ctx)
{
    if (!ctx) return WF_OK;
    vit_turn my_turn(ctx->iter_lock);
    return vit_quiesce_locked(ctx);
}

extern "C" int wf_viterbi4_iteration_host(wf_ctx *ctx, void *d_state, int length, int differential,
                                          const double *h_mf3_ri, double *h_bits_out, double *h_syms_out, void *stream)
{
    WF_REQUIRE(ctx && d_state && h_mf3_ri && h_bits_out && h_syms_out, "wf_viterbi4_iteration_host: NULL argument");
    WF_REQUIRE(length >= 1 && length <= VIT_MAX_LEN, "wf_viterbi4_iteration_host: length %d", length);
    WF_HIP(hipSetDevice(ctx->device));
    const bool use_server = ctx->opt[WF_OPT_ITERATION_SERVER] == 0;
    hipStream_t s = wf_stream(stream);
    if (!use_server) {      // one launch + one synchronise per call (the round-2 form; kept for A/B and as a fallback)
        if (!ctx->h_iter) {
            WF_HIP(hipHostMalloc(&ctx->h_iter, (6 + 2 * VIT_MAX_LEN) * sizeof(double), hipHostMallocMapped));
            WF_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->d_iter), ctx->h_iter, 0));
        }
        for (int k = 0; k < 6; ++k) ctx->h_iter[k] = h_mf3_ri[k];
        hipLaunchKernelGGL(viterbi_iteration_kernel, dim3(1), dim3(64), 0, s, static_cast<vit_state *>(d_state), length,
                           differential ? 1 : 0, ctx->d_iter, ctx->d_iter + 6, ctx->d_iter + 6 + VIT_MAX_LEN);
        WF_LAUNCH_CHECK();
        WF_HIP(hipStreamSynchronize(s));
        bool undefined = false;
        for (int k = 0; k < length; ++k) {
            h_bits_out[k] = ctx->h_iter[6 + k];
            h_syms_out[k] = ctx->h_iter[6 + VIT_MAX_LEN + k];
            undefined = undefined || h_bits_out[k] != h_bits_out[k];
        }
        if (undefined) {        // as on the server path below: the reference raises KeyError here (model.py:171-174)
            wf_set_error("traceback reached a state pair with no connecting branch");
            return WF_ERR_KEY;
        }
        return WF_OK;
    }
    vit_turn my_turn(ctx->iter_lock);      // one request at a time
    WF_REQUIRE(!ctx->iter_closing, "wf_viterbi4_iteration_host: the context has been retired (interpreter exit / wf_ctx_retire)");
    if (!ctx->h_mailbox) {
        WF_HIP(hipHostMalloc(&ctx->h_mailbox, sizeof(vit_mailbox), hipHostMallocMapped | hipHostMallocCoherent));
        memset(ctx->h_mailbox, 0, sizeof(vit_mailbox));
        WF_HIP(hipHostGetDevicePointer(&ctx->d_mailbox, ctx->h_mailbox, 0));
        hipStream_t is;
        WF_HIP(hipStreamCreateWithFlags(&is, hipStreamNonBlocking));
        ctx->iter_stream = is;
    }
    vit_mailbox *mb = static_cast<vit_mailbox *>(ctx->h_mailbox);
    if (ctx->iter_last_state != d_state) {
        // a detector this server has not seen: whatever the caller queued on ITS stream to set the state up
        // (the zero fill of a new detector) must have landed before the server touches it
        WF_HIP(hipStreamSynchronize(s));
        ctx->iter_last_state = d_state;
    }
    const unsigned long long seq = mb->req + 1;
    mb->req = seq;
    unsigned long long payload[8];
    payload[0] = (unsigned long long)(uintptr_t)d_state;
    payload[1] = (unsigned long long)(unsigned)length | ((unsigned long long)(differential ? 1u : 0u) << 32);
    memcpy(&payload[2], h_mf3_ri, 6 * sizeof(double));
    for (int k = 0; k < 8; ++k) wf_store16(&mb->chunk[k][0], payload[k], seq);      // one 16-byte store per chunk: never torn
    const int nans = 2 * length;
    const double t_post = vit_now_s();
    for (unsigned spins = 0;; ++spins) {
        // the last chunk first (the device writes them low to high), then every tag
        if (__atomic_load_n(&mb->ans[nans - 1][1], __ATOMIC_ACQUIRE) == seq) {
            bool all = true;
            for (int k = 0; k < nans; ++k) all = all && __atomic_load_n(&mb->ans[k][1], __ATOMIC_ACQUIRE) == seq;
            if (all) break;
        }
        if (__atomic_load_n(&mb->running, __ATOMIC_ACQUIRE) == 0) {
            // not started yet, or retired after its idle time (possibly while this request was being posted):
            // (re)start it — it picks the pending request up from the sequence numbers
            __atomic_store_n(&mb->running, 1ull, __ATOMIC_RELEASE);
            hipLaunchKernelGGL(viterbi_iteration_server_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(ctx->iter_stream),
                               static_cast<vit_mailbox *>(ctx->d_mailbox));
            WF_LAUNCH_CHECK();
        }
        if ((spins & 0xFFFFu) == 0xFFFFu && vit_now_s() - t_post > VIT_HOST_SPIN_SECONDS) {   // the device is gone or wedged
            wf_set_error("wf_viterbi4_iteration_host: the iteration server did not answer");
            return WF_ERR_HIP;
        }
        __builtin_ia32_pause();
    }
    bool undefined = false;
    for (int k = 0; k < length; ++k) {
        // (value and tag of a chunk arrive in one 16-byte write; the value is read after its tag matched)
        const unsigned long long vb = __atomic_load_n(&mb->ans[k][0], __ATOMIC_RELAXED), vs = __atomic_load_n(&mb->ans[length + k][0], __ATOMIC_RELAXED);
        memcpy(&h_bits_out[k], &vb, 8);
        memcpy(&h_syms_out[k], &vs, 8);
        undefined = undefined || h_bits_out[k] != h_bits_out[k];
    }
    if (undefined) {        // the reference's reverse_transitions lookup raises KeyError here (model.py:171-174); outputs hold NaN
        wf_set_error("traceback reached a state pair with no connecting branch");
        return WF_ERR_KEY;
    }
    return WF_OK;
}
