// wf_cpm_lanes.hip — the generic CPM trellis detector with one LANE per chunk (round 4).
//
// wf_cpm_detect.hip maps a trellis of <= 16 states onto one 16-lane DPP row (lane = state): every call is a chain of
// three dependent LDS round trips (rotation, candidate exchange, survivor fetch) plus a four-step DPP all-reduce, and
// four chunks share a wave — 0.42-0.44 of vector issue with the LDS pipe half busy (profiles/r03_pmc_cpm_viterbi_*).
// Here a LANE runs the whole sequential detector of its chunk: the S metrics, tilted phase indices and decision
// registers of the trellis live in that lane's registers, the trellis permutation is compile-time arithmetic (every array
// index below is a constant expression, so nothing is indexed dynamically and nothing goes to scratch), the 16 states
// of a call are independent instruction streams (no cross-lane traffic, no exchange, no wave barrier), and a wave
// advances 64 chunks per instruction.  Vector instructions per call and chunk: 10.3 (ARTM, 16 states x 4 branches)
// and 3.5 (PCM/FM, 10 states x 2) against 15.5 and 10.3 in the row form.
//
// Rows reach the lanes through a per-wave LDS ring filled by LDS-DMA (global_load_lds_dwordx4): one instruction moves
// 1 KiB = 128 B (8 pieces of 16 B: half an ARTM call, two PCM/FM calls) of eight consecutive chunks, so HBM sees whole
// 128 B lines, never a lane-strided gather.  A ring slot is 8 KB (8 such instructions: the 64 chunks of the wave); the
// ring image is lane-linear (the DMA's destination is M0 + lane * 16), so the bank swizzle is applied on the SOURCE
// side: place j of chunk c holds piece j ^ ((c >> 1) & 7), and lane c reads piece q at place q ^ ((c >> 1) & 7) —
// conflict-free ds_read_b128 (the 16 lanes of a service group hit 16 different 4-bank groups).  The DMAs are inline asm
// (hipcc drains a builtin LDS-DMA with vmcnt(0) before the next ds_read; MI355X guide, "Pipelining across barriers"):
// R slots in flight per wave, retired by a counted s_waitcnt vmcnt(8 (R - 1)); the ring is private to its wave, so no
// barrier is involved.  The ring is small on purpose — R = 3: 27 KB, R = 2: 18.5 KB — because in the pipelined links this kernel runs
// BESIDE the next block's front end, whose workgroups take 40 KB each: a lane wave has to fit the LDS one of them frees.
//
// Arithmetic, tie-breaks and emission are those of the row form and of cpm_oracle.c (the sequential statement kept
// with the tests), operation for operation: inc = -fma(cos, Re z, sin * Im z), candidate = metric + inc, strict '<'
// in list order (start state ascending, then input ascending), metrics minus their minimum, first state whose
// normalised metric is 0.0 emits bits [lgM (D - 1) ..] of its decision register.  The proof records (start / end state
// per chunk) have the row form's layout: cpm_verify_kernel and cpm_repair_kernel (wf_cpm_detect.hip) serve both.
#include <stddef.h>
#include <stdlib.h>

#include <type_traits>

#include "wf_cpm_detect.h"

#define LANE_SLOT_PIECES 8              // 16-byte pieces of a chunk per ring slot (one 128-byte line)
#define LANE_SLOT_BYTES 8192            // one ring slot: 64 chunks x 8 pieces x 16 B
#define LANE_DMAS 8                     // LDS-DMA instructions per slot (1 KiB each)
#define LANE_ROT_GAP 2056               // bytes between the cos and the sin column of the rotation table
#define LANE_ROT_BYTES (LANE_ROT_GAP + 64 * 8)
#define LANE_LDS_BYTES(R) ((R) * LANE_SLOT_BYTES + LANE_ROT_BYTES)

template <int M_, int LP_, int NC_, int P_, int NH_, int K0_, int K1_>
struct lane_spec {
    static constexpr int M = M_, LP = LP_, NC = NC_, P = P_, NH = NH_, K0 = K0_, K1 = K1_;
    static constexpr int LGM = M == 4 ? 2 : 1;
    static constexpr int ipow(int b, int e)
    {
        int r = 1;
        for (int i = 0; i < e; ++i) r *= b;
        return r;
    }
    static constexpr int NCORR = ipow(M, LP - 1), MSUB = ipow(M, LP - 2), NF = ipow(M, LP), S = NC * NCORR;
    // calls software-pipelined (see `step`): pays while the carried operands fit the register file — same box, detector
    // stage of the link, off | on: PCM/FM (10 states) 0.339 / 0.345 | 0.290 / 0.297 ms; ARTM (16 states x 4 branches, 240
    // registers without it) 0.688 / 0.715 | 0.784 / 0.793 (profiles/r04_ab_lane_switches.log)
    static constexpr bool PIPE = S * M <= 32;
    static_assert((M == 2 || M == 4) && LP >= 2 && S <= 16 && NF >= 4 && NF <= 16 && P % NC == 0 && 2 * P <= 64,
                  "lane form: trellis of <= 16 states, 4 .. 16 filters per call, pulse of >= 2 symbols");
};

template <int I, int N, class F>
__device__ __forceinline__ void lane_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        lane_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ double lane_min(double a, double b)      // v_min_f64 as the row form issues it (no NaNs to quieten)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// c ? a : b as ONE v_cndmask_b32 the optimiser cannot see through.  Written as a plain select chain over the S decision
// registers, the emission below was turned into "store the registers to scratch, select an offset, load": a
// scratch_load in the call loop — and the compiler's vmcnt(0) for it drained the row prefetch on every call.
__device__ __forceinline__ uint32_t lane_sel(bool c, uint32_t a, uint32_t b)
{
    uint32_t r;
    const uint64_t mk = __builtin_amdgcn_ballot_w64(c);
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mk));
    return r;
}

template <int LO, int HI, int S>
__device__ __forceinline__ double lane_tree_min(const double (&v)[S])
{
    if constexpr (HI - LO == 1) return v[LO];
    else return lane_min(lane_tree_min<LO, (LO + HI) / 2, S>(v), lane_tree_min<(LO + HI) / 2, HI, S>(v));
}

__device__ __forceinline__ int64_t lane_uniform64(int64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Four LDS-DMA instructions: 16 B per lane from sbase + voff (bytes) to LDS lds, lds + 1 KiB, ... (M0 = destination;
// written and restored inside the statement — the compiler owns M0 everywhere else).  The instruction's immediate offset
// moves the LDS destination AND the global source (tools/lds_dma_offset_probe.hip: offset:1024 with M0 = 0 puts global
// byte 1024 at LDS byte 1024), so one M0 serves the four: the caller passes voff_k - 1024 k (+ LANE_DMA_BIAS on the
// offsets, - LANE_DMA_BIAS on the base: offsets are unsigned).  Rows are read once: nt.
#define LANE_DMA_BIAS 4096
__device__ __forceinline__ void lane_dma4(unsigned v0, unsigned v1, unsigned v2, unsigned v3, const void *sbase_biased, unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5 nt\n\t"
                 "global_load_lds_dwordx4 %2, %5 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %4, %5 offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase_biased), "s"(lds)
                 : "memory");
}

// One LDS-DMA instruction with a full per-lane address (the slots at either end of a burst, whose rows are clamped).
__device__ __forceinline__ void lane_dma1(const void *src, unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds)
                 : "memory");
}

// A 32-bit value widened to 64 bits INSIDE a cold branch of the call loop (the proof records written at a chunk's first and last
// call): the value is made opaque where it is used, so the widening — a v_mov of zero per record word — cannot be speculated
// out of the branch into every call (it was: ~70 moves per call, 6 % of the loop's vector instructions, for stores that run twice per chunk).
__device__ __forceinline__ uint64_t lane_cold_u64(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return (uint64_t)v;
}

template <int N>
__device__ __forceinline__ void lane_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct cpm_lane_params {
    int64_t ncalls, nchunks;
    int64_t slack_lo, slack_hi;     // rows of addressable memory before / behind the row array (0: none promised; MF form: groups of 8 samples)
    int CH, W, D;
    // MF form (the matched filters run in the lane: `rows` are the noisy SAMPLES, call k's window = samples 8 k .. 8 k + 8)
    const double *templ;            // templates [nh][NF][9] complex, f <-> NF - 1 - f exact conjugates (the caller checked)
    int64_t nsamp;                  // samples addressable from `rows`
    int col0;                       // call k takes template column (k + col0) % nh
};
typedef const __attribute__((address_space(4))) double *lane_cdp;    // (read through the scalar cache: the templates are constants of the launch)

// DHI: the emitted field (bits lgM (D - 1) ... of the best state's decision register) sits in the register's high half.
// SOLO: the kernel claims accumulation registers it never uses (288 registers in all), so that the dispatcher cannot put two
// of its waves on one SIMD.  Why: inside a link the 611 one-wave workgroups of a launch do NOT land one per SIMD as they do in
// a fresh process (tools/placement_probe.hip) — time stamps of a launch showed 9 SIMDs with two waves while 400 stood empty,
// and the launch lasting 37 % longer than its typical wave (profiles/r05_lane_wave_lifetimes.log).  Alone the claim is worth
// 7 - 15 % of the detector (profiles/r05_ab_lane_solo.log); beside a front end, whose waves free 128 registers at a time, a
// wave that needs 288 waits longer (PCM/FM link 0.608 -> 0.653 ms): the pipelined links launch the plain instantiation.
//
// MF (round 6): the lane runs the MATCHED FILTERS too.  `rows` are then the noisy samples (call k's window: samples 8 k .. 8 k + 8,
// 128 new bytes per call where a row of 16 filter outputs is 256), a ring slot is exactly one call's 8 new samples of the wave's 64
// chunks, and the 16 filter outputs of a call are formed in registers from the templates' conjugate pairs, T[15 - p] == conj(T[p]):
// with r = x + j y and T_p = c + j s, four real 9-tap chains  P = sum x c,  Q = sum y s,  R = sum y c,  U = sum x (-s)  (each
// acc = fma(sample, tap, acc) from +0.0, k ascending: the order v_mfma_f64_16x16x4_f64 runs, i.e. bit for bit what the paired
// front end mod_chan_bank_kernel<.., 32> stores) give  Z_p = (P + Q) + j (U + R)  and  Z_{15-p} = (P - Q) - j (U - R).
// 288 multiply-adds per call and lane against 128 B per call and lane that never cross HBM; the taps are scalar operands
// (s_load through the constant cache, 18 doubles per pair), so a multiply-add costs its one vector instruction.
template <class SP, int R, bool DHI, bool SOLO, bool MF = false>
__global__ __launch_bounds__(64, 1) void cpm_lane_kernel(const double2 *__restrict__ rows, const double2 *__restrict__ rot_cs,
                                                         uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                         uint64_t *__restrict__ edge, cpm_lane_params P)
{
    constexpr int S = SP::S, M = SP::M, NF = SP::NF, LGM = SP::LGM, NC = SP::NC, TWO_P = 2 * SP::P;
    constexpr int UNR = 2;                                           // calls per loop trip: the leaving symbol's parity and every piece's place in its slot are constants
    constexpr int PPC = MF ? 8 : NF;                                 // 16-byte pieces of a chunk's stream per call: its filter outputs | its 8 new samples
    static_assert((UNR * PPC) % LANE_SLOT_PIECES == 0 && LANE_SLOT_PIECES % M == 0, "a loop trip takes whole slots, a filter group sits inside one");
    static_assert(LANE_DMAS * (R - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert(!MF || (!SP::PIPE && NF == 16 && M == 4 && SP::NCORR == 4 && R >= 2), "MF form: the 16-filter, 4-group trellis (ARTM)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
#ifndef LANE_MF_PRIO
#define LANE_MF_PRIO 0
#endif
    if constexpr (MF && !SOLO && LANE_MF_PRIO > 0) __builtin_amdgcn_s_setprio(LANE_MF_PRIO);   // (beside a front end: see the A/B at LANE_MF_PRIO's use in DESIGN)
    if constexpr (SOLO) {
        if constexpr (SP::M == 4) asm volatile("; accumulation registers claimed, never used: one lane wave per SIMD" ::: "a63");
        else asm volatile("; accumulation registers claimed, never used: one lane wave per SIMD" ::: "a127");
    }
    // Rotation table as two 8-byte columns read by two ds_read_b64 (64-bank mode: the 2p <= 64 entries of a column sit in
    // distinct banks, equal entries are broadcast).  One ds_read2st64_b64 — what the compiler makes of columns 1 KB
    // apart — is served in 32-bank mode, where entries 16 apart collide: 40 % of this kernel's LDS cycles were bank
    // conflicts (profiles/r04_pmc_lane_first.json).  LANE_ROT_GAP is neither a ds_read2_b64 nor a ds_read2st64_b64 distance.
    char *const rot_cos = smem + R * LANE_SLOT_BYTES;
    char *const rot_sin = smem + R * LANE_SLOT_BYTES + LANE_ROT_GAP;
    for (int k = lane; k < TWO_P; k += 64) {
        const double2 e = rot_cs[k];
        reinterpret_cast<double *>(rot_cos)[k] = e.x;
        reinterpret_cast<double *>(rot_sin)[k] = e.y;
    }
    if (blockIdx.x == 0 && lane < CPM_NLIST) cpm_list_counts(edge, P.nchunks, CPM_EDGE_WORDS)[lane] = 0;   // the repair lists (wf_cpm_detect.h): empty
    const int64_t n0 = state ? lane_uniform64((int64_t)state[CPM_ST_N]) : 0;    // calls made before this launch
    const int64_t chunk0 = (int64_t)blockIdx.x * 64;
    const int64_t chunk = chunk0 + lane;
    const int64_t k_first = chunk * P.CH;
    const bool live = k_first < P.ncalls;
    // The step is unrolled with the leaving symbol's parity as a constant: the wave starts its warm-up one call
    // earlier when that makes the parity of its first call 0 (chunk starts and chunk lengths are even).
    const int Weff = P.W + (SP::NH == 2 ? (int)((n0 - P.W - SP::LP + 1) & 1) : 0);
    const int T = __builtin_amdgcn_readfirstlane(Weff + P.CH);        // calls a lane runs (uniform, and said so: see fetch)
    const int nslots = MF ? T + 1 : (int)(((int64_t)T * NF + LANE_SLOT_PIECES - 1) / LANE_SLOT_PIECES);   // (MF: the last call's ninth sample is piece 0 of one more slot)
    const int dshift = LGM * (P.D - 1) - (DHI ? 32 : 0);            // inside its 32-bit half

    // Detector state of this chunk.  (Tried: the decision registers in LDS, [state][lane], the winner's fetched by
    // address from ONE selected word — rotation offset | register address — instead of two more v_cndmask_b32 per
    // candidate: 14 % fewer vector instructions, 30 registers less, and 7 % (ARTM) / 22 % (PCM/FM) SLOWER on the same
    // box: 623 -> 666 us and 236 -> 288 us; the register form stays.)
    double m[S];
    uint32_t r8[S];                                                  // 8 r: byte offset of the survivor's rotation in a table column
    uint64_t h[S];
    constexpr uint32_t TWO_P8 = 8u * TWO_P;
    {
        const int64_t k_start = chunk == 0 ? 0 : k_first - Weff;     // first call this lane really runs
        const int tilt = cpm_tilt(M, SP::P, SP::NH, SP::K0, SP::K1, SP::LP, n0 + k_start);
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            m[s] = 0.0;
            const int v = 2 * (s % NC) - tilt;
            r8[s] = 8u * (uint32_t)(v < 0 ? v + TWO_P : v);
            h[s] = 0;
        });
    }
    if (state && chunk == 0 && n0 > 0) {                            // continue the carried detector
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            m[s] = __longlong_as_double((long long)state[CPM_ST_M + s]);
            r8[s] = 8u * (uint32_t)state[CPM_ST_V + s];
            h[s] = state[CPM_ST_H + s];
        });
    }
    uint64_t *const erec = edge + chunk * CPM_EDGE_WORDS;            // (written only when the chunk is live)
    uint64_t acc = 0;                                                // decisions of the current group of 8 calls, one byte each

    // What a group of start states (one `corr`) needs: its M filter outputs and the NC rotations of its survivors.
    struct operands {
        double2 z[M];
        double cr[NC], sr[NC];
    };
    auto load_rot = [&](auto cc, operands &o) __attribute__((always_inline)) {
        lane_for<0, NC>([&](auto lc) {
            constexpr int src = decltype(lc)::value + NC * decltype(cc)::value;
            o.cr[decltype(lc)::value] = *reinterpret_cast<const double *>(rot_cos + r8[src]);
            o.sr[decltype(lc)::value] = *reinterpret_cast<const double *>(rot_sin + r8[src]);
        });
    };
    operands first;     // group 0 of the call about to run, requested by the call before it (pipelined calls only)

    // One detector call.  KV: the leaving symbol's variant (compile-time); zsrc(f): filter output f of this call.
    // PIPE: group 0's operands were requested by the previous call (`first`), and this call requests the next one's —
    // znext(f): filter output f of the NEXT call — before its own tail (normalisation, decision registers, emission):
    // the tail's ~80 vector instructions hide that round trip, and the slot hand-over (DMA issue + counted wait) that
    // znext may carry sits there too, instead of in front of a call whose first group then waits for its operands
    // (lane_spec::PIPE).
    // MF: zprep(0) leaves the filter outputs of groups 0 and 3 where zsrc finds them, zprep(1) — behind group 0's work — those of
    // groups 1 and 2 (a conjugate pair of filters sits in groups g and 3 - g): at most two groups' outputs wait in registers.
    auto step = [&](auto kvc, auto pipec, auto &&zsrc, auto &&znext, auto &&zprep, bool emit, bool emit_ok, int group_pos) __attribute__((always_inline)) {
        constexpr int KV = decltype(kvc)::value;
        constexpr bool PIPE = decltype(pipec)::value;
        constexpr int K_old = KV == 2 ? 0 : (KV == 1 ? SP::K1 : SP::K0);
        double nm[S];
        uint32_t nr8[S];
        uint64_t nh[S];
        // Start states grouped by `corr` (their Lp - 1 previous symbols): a group needs M filter outputs and NC rotations
        // and hands exactly one candidate to each of NC * M end states — in the order of the sequential statement's branch list (start
        // state ascending), so the running strict '<' below IS its first arg-min: the first listed branch keeps a tie.
        // (Grouped by end state instead, all S rotations and all M^Lp filter outputs stay live across the call: 354
        // registers for ARTM, a third of them parked in AGPRs, and as many compare masks as the SGPR file holds.)
        // A lane is alone on its SIMD for most of a burst (1e7 calls make ~600 waves for 1024 SIMDs), so nothing but
        // its own instruction stream hides an LDS round trip: the operands of group g + 1 are requested BEFORE group g
        // is worked (profiles/r04_pmc_lane_first.json: 29 % of the wave's cycles sat in s_waitcnt lgkmcnt without it).
        auto load_group = [&](auto cc, operands &o) __attribute__((always_inline)) {
            constexpr int corr = decltype(cc)::value;
            lane_for<0, M>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                o.z[u] = zsrc(std::integral_constant<int, u + M * corr>{});
            });
            load_rot(cc, o);
        };
        auto run_group = [&](auto self, auto cc, const operands &o) __attribute__((always_inline)) -> void {
            constexpr int corr = decltype(cc)::value;
            constexpr int u_old = corr / SP::MSUB;                   // the symbol leaving the window: slot index of these candidates
            constexpr int inc = (K_old * u_old) % SP::P;
            constexpr uint32_t delta8 = 8u * (uint32_t)(((2 * inc - (M - 1) * K_old) % TWO_P + TWO_P) % TWO_P);   // what the branch adds to the TILTED phase index
            operands nx;
            if constexpr (corr + 1 < SP::NCORR) {
                if constexpr (MF) load_rot(std::integral_constant<int, corr + 1>{}, nx);     // (the filter outputs come from registers: below)
                else load_group(std::integral_constant<int, corr + 1>{}, nx);
            }
            lane_for<0, NC>([&](auto lc) {
                constexpr int cls = decltype(lc)::value;
                constexpr int src = cls + NC * corr;
                const double cr = o.cr[cls], sr = o.sr[cls];
                const uint32_t x = r8[src] + delta8;                                        // < 4p (x 8)
                const uint32_t rs = min(x, x - TWO_P8);                                    // mod 2p: the difference wraps to a huge value when x < 2p
                lane_for<0, M>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    constexpr int e = (cls + inc) % NC + NC * (u + M * (corr % SP::MSUB));
                    const double inc_m = -fma(cr, o.z[u].x, sr * o.z[u].y);              // -Re(e^{-j theta} Z)
                    const double c = m[src] + inc_m;
                    if constexpr (u_old == 0) {
                        nm[e] = c;
                        nr8[e] = rs;
                        nh[e] = h[src];
                    } else {
                        const bool f = c < nm[e];
                        nm[e] = lane_min(nm[e], c);
                        nr8[e] = f ? rs : nr8[e];
                        nh[e] = f ? h[src] : nh[e];
                        // (pins the phase select next to its compare: nothing needs it before the end of the call, so the
                        // optimiser sank all 48 of them — and their 48 lane masks, the whole scalar file — below the emission.
                        // Pinning the decision register's two selects as well: 8 scalar spills instead of 12, and 4 % slower
                        // — profiles/r04_ab_lane_switches.log)
                        asm volatile("" : "+v"(nr8[e]));
                    }
                });
                // (the M candidates of a start state end here for the instruction scheduler: left free, it lines up all
                // of a call's compares first and keeps their lane masks — two scalar registers each — alive until the
                // selects: 100+ scalar spills, reloaded by v_readlane inside the call loop)
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (MF && corr + 1 < SP::NCORR) {
                if constexpr (corr == 0) zprep(std::integral_constant<int, 1>{});
                lane_for<0, M>([&](auto uc) { nx.z[decltype(uc)::value] = zsrc(std::integral_constant<int, decltype(uc)::value + M * (corr + 1)>{}); });
            }
            if constexpr (corr + 1 < SP::NCORR) self(self, std::integral_constant<int, corr + 1>{}, nx);
        };
        zprep(std::integral_constant<int, 0>{});
        if constexpr (PIPE) {
            const operands o0 = first;
            run_group(run_group, std::integral_constant<int, 0>{}, o0);
        } else {
            operands o0;
            load_group(std::integral_constant<int, 0>{}, o0);
            run_group(run_group, std::integral_constant<int, 0>{}, o0);
        }
        lane_for<0, S>([&](auto sc) { r8[decltype(sc)::value] = nr8[decltype(sc)::value]; });
        if constexpr (PIPE) {                                        // the next call's first group: its survivors' phases are known now
            lane_for<0, M>([&](auto uc) { first.z[decltype(uc)::value] = znext(uc); });
            load_rot(std::integral_constant<int, 0>{}, first);
        }
        const double gmin = lane_tree_min<0, S, S>(nm);
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            m[s] = nm[s] - gmin;                                     // the minimum becomes exactly 0.0
            h[s] = (nh[s] << LGM) | (uint64_t)((s / NC) % M);        // the newest symbol of every branch into s
        });
        if (emit) {
            // np.argmin: the FIRST state whose metric is the minimum emits — walk down so that the lowest index wins.
            // Only one 32-bit half of the register holds the field (lgM (D - 1) is even for lgM = 2): DHI.
            auto half = [](uint64_t v) __attribute__((always_inline)) { return DHI ? (uint32_t)(v >> 32) : (uint32_t)v; };
            uint32_t w = half(h[S - 1]);
            lane_for<1, S>([&](auto qc) {
                constexpr int s = S - 1 - decltype(qc)::value;
                w = lane_sel(m[s] == 0.0, half(h[s]), w);
                if constexpr (s % 4 == 0) __builtin_amdgcn_sched_barrier(0);       // (at most four lane masks alive at a time)
            });
            uint32_t sym = (w >> dshift) & (uint32_t)(M - 1);
            sym = emit_ok ? sym : 0u;                                // calls before the D-th of the burst decide nothing
            acc |= (uint64_t)sym << (8 * group_pos);
        }
    };
    using kv2 = std::integral_constant<int, 2>;
    // ---- MF: the call's window (9 samples, planes x | y) and its 16 filter outputs
    double wx[MF ? 9 : 1], wy[MF ? 9 : 1];
    double2 zz[MF ? NF : 1];
    // template column of the calls a loop trip runs first / second (uniform): local call k takes column (k + col0) % nh,
    // k = chunk CH - Weff + t and chunks start on even calls
    lane_cdp tcol0 = nullptr, tcol1 = nullptr;
    if constexpr (MF) {
        const int flip = SP::NH == 2 ? __builtin_amdgcn_readfirstlane((Weff + P.col0) & 1) : 0;
        const lane_cdp t0 = (lane_cdp)(uintptr_t)P.templ;
        tcol0 = t0 + (flip ? 2 * NF * 9 : 0);
        tcol1 = SP::NH == 2 ? t0 + (flip ? 0 : 2 * NF * 9) : t0;
    }
    // pairs 4 ph .. 4 ph + 3 of the window in wx / wy against column tc: zz[p] and zz[15 - p] (groups ph and 3 - ph)
    auto mf_pairs = [&](auto phc, lane_cdp tc) __attribute__((always_inline)) {
        asm volatile("" : "+s"(tc));                                 // (re-read per call through the constant cache, not 288 loop-invariant scalar registers)
        lane_for<0, 4>([&](auto pc) {
            constexpr int p = 4 * decltype(phc)::value + decltype(pc)::value;
            double sP = 0.0, sQ = 0.0, sR = 0.0, sU = 0.0;
            lane_for<0, 9>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double c = tc[2 * (p * 9 + k)], sn = tc[2 * (p * 9 + k) + 1];
                sP = fma(wx[MF ? k : 0], c, sP);
                sU = fma(wx[MF ? k : 0], -sn, sU);
                sQ = fma(wy[MF ? k : 0], sn, sQ);
                sR = fma(wy[MF ? k : 0], c, sR);
            });
            zz[MF ? p : 0] = make_double2(sP + sQ, sU + sR);
            zz[MF ? NF - 1 - p : 0] = make_double2(sP - sQ, -(sU - sR));
        });
    };
    auto zreg = [&](auto fc) __attribute__((always_inline)) { return zz[MF ? decltype(fc)::value : 0]; };
    auto noprep = [](auto) __attribute__((always_inline)) {};

    // EVERY lane runs EVERY call of the loop below — the row fetch inside a call is the whole wave's business (a lane
    // fetches pieces of other lanes' chunks), so no call may sit in divergent control flow.  Lanes whose call is not
    // theirs to run compute on whatever the clamped rows hold, and what must not see that is kept out of its way:
    //   - chunk 0 has no warm-up: its lane is handed its true state when its first own call comes up (t == t_first0);
    //   - a chunk ends where the burst ends: its end record and the carry are written when its last call is done
    //     (t + 1 == t_hi), not after the loop; decisions past t_hi are never stored;
    //   - lanes without a chunk fetch a live chunk's rows again and write nothing.
    // Virtual pre-start symbols: the first LP - 1 calls of a fresh burst have no symbol leaving the window (variant 2).
    // Only chunk 0 meets them; it runs them here, straight from global memory, and skips them in the loop.
    const int64_t kmin0 = n0 < SP::LP - 1 ? (SP::LP - 1 - n0 < P.ncalls ? SP::LP - 1 - n0 : P.ncalls) : 0;     // (uniform)
    if (chunk == 0) {
        for (int64_t k = 0; k < kmin0; ++k) {
            if constexpr (MF) {
                lane_for<0, 9>([&](auto jc) {
                    const double2 v = rows[8 * k + decltype(jc)::value];
                    wx[decltype(jc)::value] = v.x;
                    wy[decltype(jc)::value] = v.y;
                });
                const lane_cdp tc = (lane_cdp)(uintptr_t)P.templ + (SP::NH == 2 && ((k + P.col0) & 1) ? 2 * NF * 9 : 0);
                mf_pairs(std::integral_constant<int, 0>{}, tc);
                mf_pairs(std::integral_constant<int, 1>{}, tc);
                step(kv2{}, std::false_type{}, zreg, zreg, noprep, true, n0 + k >= P.D - 1, (int)(k & 7));
            } else {
                const double2 *zr = rows + k * NF;
                auto zg = [&](auto fc) { return zr[decltype(fc)::value]; };
                step(kv2{}, std::false_type{}, zg, zg, noprep, true, n0 + k >= P.D - 1, (int)(k & 7));
            }
        }
        // parked in chunk 0's own start record (nothing else reads that one) until the loop reaches its first call
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            erec[3 * s] = (uint64_t)__double_as_longlong(m[s]);
            erec[3 * s + 1] = (uint64_t)r8[s];
            erec[3 * s + 2] = h[s];
        });
    }
    // Call t of the loop is local call kbase + t.  32-bit bounds on t (64-bit compares per call cost the loop its scalar
    // registers): t_hi = end of what this lane owns (0: nothing), t_ok = first call whose decision counts.
    const int64_t kbase = k_first - Weff;
    auto clamp_t = [&](int64_t v) __attribute__((always_inline)) { return (int)(v < 0 ? 0 : (v > T ? T : v)); };
    const int t_first0 = Weff + (int)kmin0;                          // (uniform) chunk 0's first call inside the loop
    const int t_hi = live ? clamp_t(P.ncalls - kbase) : 0;
    const int t_ok = max(clamp_t((int64_t)P.D - 1 - n0 - kbase), chunk == 0 ? t_first0 : 0);
    uint8_t *const outp = out + kbase;                               // out[k] = outp[t]
    const bool owns_last = state && live && k_first + P.CH >= P.ncalls;

    // ---- the ring.  A lane's rows are one stream of 16-byte pieces, NF per call; slot s of the ring holds pieces
    // [8 s, 8 s + 8) of every chunk of the wave.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int c8 = lane >> 3, j8 = lane & 7;
    // The DMAs take a uniform base + a per-lane offset, nothing per lane is clamped.  That is right while every row the
    // wave asks for exists: from slot fast_lo (the rows of chunk 0's warm-up lie before the burst) to slot fast_hi (a
    // burst ends inside, or short of, the last wave's chunks) — outside that range, in the wave at either end of a
    // burst, every DMA works out and clamps its own row.  (Both waves went through the careful form for ALL their
    // slots at first: 35 % more instructions per call, and a launch lasts as long as its slowest wave.)  Lanes
    // without a chunk fetch the wave's last live chunk again.  slack_lo / slack_hi: rows of addressable memory the
    // caller vouches for before and behind the array (the link's workspace has them: no careful slot at all).
    const int64_t cg_last = chunk0 + 63 < P.nchunks - 1 ? chunk0 + 63 : P.nchunks - 1;     // last live chunk of the wave
    int fast_lo, fast_hi;
    {
        // slot s reads rows floor(8 s / NF) .. floor((8 s + 7) / NF) of a chunk's stream, i.e. absolute rows kbase(chunk) + those
        // (MF: a stream "row" is a call's 8 new samples, and the array holds nsamp / 8 whole ones)
        const int64_t nrows_tot = MF ? P.nsamp / 8 : P.ncalls;
        const int64_t need_lo = Weff - chunk0 * P.CH - P.slack_lo;                       // first stream row that is >= -slack_lo for chunk0
        fast_lo = need_lo <= 0 ? 0 : (int)((need_lo * PPC + LANE_SLOT_PIECES - 1) / LANE_SLOT_PIECES);
        const int64_t room = nrows_tot + P.slack_hi + Weff - cg_last * P.CH;             // stream rows of the last live chunk that exist
        const int64_t hi = room <= 0 ? -1 : (room * PPC) / LANE_SLOT_PIECES - 1;         // last slot that lies wholly inside them
        fast_hi = hi > 0x3fffffff ? 0x3fffffff : (int)hi;
    }
    const double2 *const srow0 = rows + (chunk0 * P.CH - Weff) * PPC;    // piece 0 of the wave's first chunk's stream (may lie before the array)
    unsigned voff[LANE_DMAS];
    lane_for<0, LANE_DMAS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        int c = 8 * i + c8;                                          // chunk of the wave this lane fetches for in DMA i
        const int q = j8 ^ ((c >> 1) & 7);                           // ... and which of the slot's 8 pieces
        c = chunk0 + c <= cg_last ? c : (int)(cg_last - chunk0);
        voff[i] = ((unsigned)c * (unsigned)P.CH * PPC + (unsigned)q) * 16u + (unsigned)(LANE_DMA_BIAS - 1024 * (i & 3));   // (see lane_dma4)
    });
    auto fetch = [&](int s, unsigned lds) __attribute__((always_inline)) {
        const int ss = s < nslots ? s : nslots - 1;                  // (past the end: the last slot again — the count of DMAs in flight must not change)
        if (ss >= fast_lo && ss <= fast_hi) {
            // (uniform by construction; said so explicitly — the ring counters are loop-carried through the call lambdas and
            //  the compiler's divergence analysis gives up on them, handing a VGPR pair to an SGPR operand)
            const char *sb = reinterpret_cast<const char *>(lane_uniform64((int64_t)(uintptr_t)(srow0 + (int64_t)ss * LANE_SLOT_PIECES))) - LANE_DMA_BIAS;
            const unsigned ld = __builtin_amdgcn_readfirstlane(lds);
            lane_dma4(voff[0], voff[1], voff[2], voff[3], sb, ld);
            lane_dma4(voff[4], voff[5], voff[6], voff[7], sb, ld + 4096);
        } else {
            // (cold, and kept small: a rolled loop that re-derives everything from values made opaque here, so that none of
            // it is unrolled into, or hoisted out of, the call loop at the cost of registers every other slot would pay for)
            int cz = c8;
            unsigned lz = __builtin_amdgcn_readfirstlane(lds);
            asm volatile("" : "+v"(cz), "+s"(lz));
#pragma unroll 1
            for (int i = 0; i < LANE_DMAS; ++i) {
                const int c = 8 * i + cz;
                const int q = j8 ^ ((c >> 1) & 7);
                const int64_t piece = (int64_t)ss * LANE_SLOT_PIECES + q;                // of the chunk's stream
                if constexpr (MF) {
                    int64_t idx = ((chunk0 + c) * P.CH - Weff) * 8 + piece;             // sample index
                    idx = idx < 0 ? 0 : (idx >= P.nsamp ? P.nsamp - 1 : idx);            // never decoded when clamped
                    lane_dma1(rows + idx, lz + 1024u * i);
                } else {
                    int64_t row = (chunk0 + c) * P.CH - Weff + piece / NF;
                    row = row < 0 ? 0 : (row >= P.ncalls ? P.ncalls - 1 : row);          // never decoded when clamped
                    lane_dma1(rows + row * NF + (piece % NF), lz + 1024u * i);
                }
            }
        }
    };
    const unsigned lds_lane = (unsigned)lane * 128u + (unsigned)((lane >> 1) & 7) * 16u;   // this chunk's 128 B of a slot, swizzle folded in

    // ring state (uniform): slot index / ring position of the next fetch and of the next slot to read
    int s_fetch = 0;
    unsigned p_fetch = 0, p_read = 0;
    auto next_pos = [](unsigned p) __attribute__((always_inline)) { return p + 1 == (unsigned)R ? 0u : p + 1; };
    for (int k = 0; k < (MF ? R : R - 1); ++k) {
        fetch(s_fetch++, lds0 + p_fetch * LANE_SLOT_BYTES);
        p_fetch = next_pos(p_fetch);
    }
    // MF: slot t holds the 8 new samples of call t; its window ends with sample 0 of slot t + 1, which is sample 0 of the next
    // call's window: carried in registers.  A call waits for slots t and t + 1, reads its 7 + 1 pieces and THEN sends the fetch of
    // slot t + R into slot t's place (the reads have returned: lgkmcnt(0)): R - 1 calls of lead for a fetch.
    double carry_x = 0.0, carry_y = 0.0;
    typedef double lane_v2d __attribute__((ext_vector_type(2)));
    if constexpr (MF) {
        lane_wait_vm<LANE_DMAS * (R - 1)>();                         // slot 0 is in
        const lane_v2d v = *reinterpret_cast<const lane_v2d *>(smem + lds_lane);
        carry_x = v.x;
        carry_y = v.y;
    }
    auto mf_window = [&]() __attribute__((always_inline)) {
        lane_wait_vm<LANE_DMAS * (R >= 2 ? R - 2 : 0)>();            // slots t and t + 1 are in (this wave's own DMAs: no barrier)
        const unsigned zb = p_read * LANE_SLOT_BYTES, zn = next_pos(p_read) * LANE_SLOT_BYTES;
        wx[0] = carry_x;
        wy[0] = carry_y;
        lane_for<1, 8>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const lane_v2d v = *reinterpret_cast<const lane_v2d *>(smem + zb + (lds_lane ^ (unsigned)(q * 16)));
            wx[MF ? q : 0] = v.x;
            wy[MF ? q : 0] = v.y;
        });
        const lane_v2d v8 = *reinterpret_cast<const lane_v2d *>(smem + zn + lds_lane);
        wx[MF ? 8 : 0] = carry_x = v8.x;
        wy[MF ? 8 : 0] = carry_y = v8.y;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the slot's pieces are in registers: its place is free
        fetch(s_fetch++, lds0 + p_read * LANE_SLOT_BYTES);
        p_read = next_pos(p_read);
    };
    unsigned zbase = 0;                                              // LDS byte offset (from smem) of the slot being read
    // the next slot: one more fetch goes out (into the position read before this one), the oldest one in flight lands
    auto acquire = [&]() __attribute__((always_inline)) {
        fetch(s_fetch++, lds0 + p_fetch * LANE_SLOT_BYTES);
        p_fetch = next_pos(p_fetch);
        lane_wait_vm<LANE_DMAS * (R - 1)>();                         // (this wave's own DMAs: no barrier)
        zbase = p_read * LANE_SLOT_BYTES;
        p_read = next_pos(p_read);
    };

    // filter output f of call UC of a loop trip (UC == UNR: call 0 of the next trip — the same places of the next slots)
    auto zring = [&](auto ucc, auto fc) __attribute__((always_inline)) {
        constexpr int piece = (decltype(ucc)::value % UNR) * NF + decltype(fc)::value;   // within the loop trip
        constexpr int q = piece % LANE_SLOT_PIECES;
        if constexpr (q == 0) acquire();                             // the first piece of a slot: hand-over
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d v = *reinterpret_cast<const v2d *>(smem + zbase + (lds_lane ^ (unsigned)(q * 16)));
        return make_double2(v.x, v.y);
    };
    if (SP::PIPE) {   // the first call's first group
        lane_for<0, M>([&](auto uc) { first.z[decltype(uc)::value] = zring(std::integral_constant<int, 0>{}, uc); });
        load_rot(std::integral_constant<int, 0>{}, first);
    }
    // (t0 is uniform, and said so on every trip: the compiler otherwise carries the counter per lane and runs the whole
    //  loop under an exec mask it keeps — and spills — in scalar registers)
    for (int t0 = 0; t0 < T; t0 = __builtin_amdgcn_readfirstlane(t0 + UNR)) {
        lane_for<0, UNR>([&](auto ucc) {
            constexpr int UC = decltype(ucc)::value;
            constexpr int KV = SP::NH == 2 ? (UC & 1) : 0;
            const int t = t0 + UC;
            if (t < T) {
                if (t == Weff && live && chunk != 0) {               // the next call is the chunk's first own one
                    lane_for<0, S>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        erec[3 * s] = (uint64_t)__double_as_longlong(m[s]);
                        erec[3 * s + 1] = lane_cold_u64(r8[s] >> 3);
                        erec[3 * s + 2] = h[s];
                    });
                }
                if (blockIdx.x == 0 && t == t_first0 && chunk == 0) {   // chunk 0: its true state, parked above
                    lane_for<0, S>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        m[s] = __longlong_as_double((long long)erec[3 * s]);
                        r8[s] = (uint32_t)erec[3 * s + 1];
                        h[s] = erec[3 * s + 2];
                    });
                    if (SP::PIPE) load_rot(std::integral_constant<int, 0>{}, first);  // (requested with the phases of the warm-up it never had)
                }
                const bool emit = t >= Weff;
                const int gp = (t - Weff) & 7;
                if constexpr (MF) {
                    step(std::integral_constant<int, KV>{}, std::false_type{}, zreg, zreg,
                         [&](auto phc) __attribute__((always_inline)) {
                             if constexpr (decltype(phc)::value == 0) mf_window();
                             mf_pairs(phc, (UC & 1) ? tcol1 : tcol0);
                         },
                         emit, t >= t_ok, gp);
                } else {
                    step(std::integral_constant<int, KV>{}, std::integral_constant<bool, SP::PIPE>{}, [&](auto fc) __attribute__((always_inline)) { return zring(ucc, fc); },
                         [&](auto fc) __attribute__((always_inline)) { return zring(std::integral_constant<int, UC + 1>{}, fc); }, noprep, emit, t >= t_ok, gp);
                }
                if (emit && gp == 7) {                               // a group of 8 decisions is complete (uniform)
                    if (t < t_hi) {
                        *reinterpret_cast<uint64_t *>(outp + (t - 7)) = acc;
                    } else if (t - 7 < t_hi) {                       // the burst ends inside the group
                        lane_for<0, 7>([&](auto qc) {                // (straight-line: a lane-dependent trip count made the whole call loop divergent)
                            constexpr int q = decltype(qc)::value;
                            if (t - 7 + q < t_hi) outp[t - 7 + q] = (uint8_t)(acc >> (8 * q));
                        });
                    }
                    acc = 0;
                }
                if (t + 1 == t_hi) {                                 // this chunk's last call: its end record, and the carry if it owns the burst's last call
                    lane_for<0, S>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        erec[48 + 3 * s] = (uint64_t)__double_as_longlong(m[s]);
                        erec[48 + 3 * s + 1] = lane_cold_u64(r8[s] >> 3);
                        erec[48 + 3 * s + 2] = h[s];
                    });
                    if (owns_last) {
                        uint64_t *const st = state;
                        st[CPM_ST_STAGE + CPM_ST_N] = (uint64_t)(n0 + P.ncalls);
                        lane_for<0, S>([&](auto sc) {
                            constexpr int s = decltype(sc)::value;
                            st[CPM_ST_STAGE + CPM_ST_M + s] = (uint64_t)__double_as_longlong(m[s]);
                            st[CPM_ST_STAGE + CPM_ST_V + s] = lane_cold_u64(r8[s] >> 3);
                            st[CPM_ST_STAGE + CPM_ST_H + s] = h[s];
                        });
                    }
                }
            }
        });
    }
    lane_wait_vm<0>();
}

// ---- the compiled specialisations: the two waveforms BASELINE configs[2] and SURVEY 8(f3) name
using lane_artm16 = lane_spec<4, 2, 4, 16, 2, 4, 5>;     // ARTM multi-h CPM, h = {4/16, 5/16}, pulse truncated to 2 symbols, 4 phase classes
using lane_pcmfm10 = lane_spec<2, 2, 5, 10, 1, 7, 7>;    // PCM/FM, h = 7/10, 5 phase classes
// Ring depth: slots of 8 KB in flight per wave (ARTM: two slots per call).  Same box, detector alone, 2 | 3 | 4 slots: ARTM 724 |
// 656 | 658 us, PCM/FM 313 | 307 | 313 us (profiles/r04_ab_lane_ring.log) — two slots leave the fetch one slot of lead, four cost a
// CU a resident wave.  In the pipelined links (what bench.py runs) a lane wave has to fit the LDS a front-end workgroup frees
// (39 KB): with two slots (18.5 KB) TWO of them do.  Same box, pipelined, 3 | 2 slots: PCM/FM 0.5688 / 0.5697 / 0.5703 | 0.5640 /
// 0.5653 / 0.5651 ms, ARTM 1.1556 / 1.1532 / 1.1548 | 1.1559 / 1.1561 / 1.1574 (profiles/r05_ab_lane_ring2_pipelined.log): two for
// PCM/FM (whose single slot holds two calls), three for ARTM.
#define LANE_R_ARTM 3
#define LANE_R_PCMFM 2
#ifndef LANE_R_ARTM_MF
#define LANE_R_ARTM_MF 3      // MF form: a slot is a whole call; a fetch has R - 1 calls of lead
#endif

// 1: no specialisation for this trellis; 0: there is one, *plan filled in.  (Which form runs is the caller's decision:
// cpm_chunk_calls, wf_cpm_detect.hip.)
int wf_cpm_lanes_plan(const wf_cpm_detector_config *d, cpm_lane_plan *plan)
{
    int spec = -1;
    if (d->M == 4 && d->Lp == 2 && d->NC == 4 && d->p == 16 && d->nh == 2 && d->K[0] == 4 && d->K[1] == 5 && d->D >= 17 && d->D <= 32) spec = 0;
    if (d->M == 2 && d->Lp == 2 && d->NC == 5 && d->p == 10 && d->nh == 1 && d->K[0] == 7 && d->D >= 1 && d->D <= 32) spec = 1;
    if (spec < 0) return 1;
    const int R = spec == 0 ? LANE_R_ARTM : LANE_R_PCMFM;
    int per_cu = (160 * 1024) / LANE_LDS_BYTES(R);
    if (per_cu > 4) per_cu = 4;                 // one wave per SIMD: 225 registers (ARTM) leave room for nothing else of this kernel
    // time per call of a chunk / of the burst (one MI355X, profiles/r04_ab_lane_ring.log, r04_lane_chunk_sweep.log):
    // ARTM 0.66 ms for 256 + 49 calls, row form 0.74 ms per 1e7; PCM/FM 0.27 ms for 256 + 64 calls, row form 0.49 ms per 1e7
    // shortest chunk (round 5, pipelined links at 1e7 symbols, same box, profiles/r05_lane_sweep.log): ARTM 192 | 256 calls at a
    // 48-call warm-up 1.181 | 1.198 ms per block; PCM/FM 192 | 256 | 320 at 64: 0.595 | 0.583 | 0.598.  (Since the repairs cascade,
    // a chunk no longer has to be long enough for a repair to meet the first trajectory inside it: this is a matter of speed only.)
    *plan = {spec, R, per_cu, spec == 0 ? 1 : 4, spec == 0 ? 192 : 256, spec == 0 ? 2160.0 : 840.0, spec == 0 ? 0.074 : 0.049};
    return 0;
}

int wf_cpm_lanes_launch(const cpm_lane_plan &plan, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri,
                        int64_t ncalls, int warmup, int chunk_calls, int64_t nchunks, uint8_t *d_decisions, void *d_state, uint64_t *d_edge,
                        void *stream, int64_t slack_lo_bytes, int64_t slack_hi_bytes, bool solo, const cpm_mf_source *mf)
{
    if (mf) {
        // the matched filters inside the lane: d_rows_ri = the noisy samples from the first call's window on
        WF_REQUIRE(plan.spec == 0, "wf_cpm_lanes: the matched-filter form serves the 16-filter ARTM design only");
        WF_REQUIRE(mf->d_templates && mf->nsamp >= 8 * ncalls + 1, "wf_cpm_lanes: %lld samples for %lld calls", (long long)mf->nsamp, (long long)ncalls);
        WF_REQUIRE(chunk_calls >= 64 && chunk_calls % 16 == 0 && warmup >= 0 && warmup % 2 == 0 && chunk_calls > warmup + 1,
                   "wf_cpm_lanes: chunk of %d calls, warm-up %d", chunk_calls, warmup);
        WF_REQUIRE((int64_t)64 * chunk_calls * 8 * 16 < (1ll << 32), "wf_cpm_lanes: chunk of %d calls overflows the 32-bit row offsets", chunk_calls);
        cpm_lane_params P{ncalls, nchunks, slack_lo_bytes / 128, slack_hi_bytes / 128, chunk_calls, warmup, det->D, mf->d_templates, mf->nsamp, mf->col0 & 1};
        const int64_t nblocks = (nchunks + 63) / 64;
        WF_REQUIRE(nblocks < (1ll << 31), "wf_cpm_lanes: burst too long for one launch");
        const size_t lds = (size_t)LANE_LDS_BYTES(LANE_R_ARTM_MF);
        using kern_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, cpm_lane_params);
        const kern_t k = solo ? static_cast<kern_t>(cpm_lane_kernel<lane_artm16, LANE_R_ARTM_MF, true, true, true>)
                              : static_cast<kern_t>(cpm_lane_kernel<lane_artm16, LANE_R_ARTM_MF, true, false, true>);
        if (lds > 48 * 1024)
            WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(64), lds, wf_stream(stream), reinterpret_cast<const double2 *>(d_rows_ri),
                           reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), d_edge, P);
        WF_LAUNCH_CHECK();
        return WF_OK;
    }
    WF_REQUIRE(chunk_calls >= 64 && chunk_calls % 16 == 0 && warmup >= 0 && warmup % 2 == 0 && chunk_calls > warmup + 1,
               "wf_cpm_lanes: chunk of %d calls, warm-up %d", chunk_calls, warmup);
    WF_REQUIRE((int64_t)64 * chunk_calls * 16 * 16 < (1ll << 32), "wf_cpm_lanes: chunk of %d calls overflows the 32-bit row offsets", chunk_calls);
    int nf = 1;
    for (int i = 0; i < det->Lp; ++i) nf *= det->M;
    cpm_lane_params P{ncalls, nchunks, slack_lo_bytes / (16 * nf), slack_hi_bytes / (16 * nf), chunk_calls, warmup, det->D, nullptr, 0, 0};
    const int64_t nblocks = (nchunks + 63) / 64;
    WF_REQUIRE(nblocks < (1ll << 31), "wf_cpm_lanes: burst too long for one launch");
    const size_t lds = (size_t)LANE_LDS_BYTES(plan.ring_batches);
    using kern_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, cpm_lane_params);
    const kern_t k = solo ? (plan.spec == 0 ? static_cast<kern_t>(cpm_lane_kernel<lane_artm16, LANE_R_ARTM, true, true>) : static_cast<kern_t>(cpm_lane_kernel<lane_pcmfm10, LANE_R_PCMFM, false, true>))
                          : (plan.spec == 0 ? static_cast<kern_t>(cpm_lane_kernel<lane_artm16, LANE_R_ARTM, true, false>) : static_cast<kern_t>(cpm_lane_kernel<lane_pcmfm10, LANE_R_PCMFM, false, false>));
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(64), lds, wf_stream(stream), reinterpret_cast<const double2 *>(d_rows_ri),
                       reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), d_edge, P);
    WF_LAUNCH_CHECK();
    return WF_OK;
}
