// wf_cpm_lanes.hip — the generic CPM trellis detector with one LANE per chunk (round 4).
//
// wf_cpm_detect.hip maps a trellis of <= 16 states onto one 16-lane DPP row (lane = state): every call is a chain of
// three dependent LDS round trips (rotation, candidate exchange, survivor fetch) plus a four-step DPP all-reduce, and
// four chunks share a wave — 0.42-0.44 of vector issue with the LDS pipe half busy (profiles/r03_pmc_cpm_viterbi_*).
// Here a LANE runs the whole sequential detector of its chunk: the S metrics, tilted phase indices and decision
// registers of the trellis live in that lane's registers, the trellis permutation is compile-time arithmetic (every array
// index below is a constant expression, so nothing is indexed dynamically and nothing goes to scratch), the 16 states
// of a call are independent instruction streams (no cross-lane traffic, no exchange, no wave barrier), and a wave
// advances 64 chunks per instruction.  Vector instructions per call and chunk: about 9.4 (ARTM, 16 states x 4
// branches) and 3.6 (PCM/FM, 10 states x 2) against 15.5 and 10.3 in the row form.
//
// Rows reach the lanes through a per-wave LDS ring filled by LDS-DMA (global_load_lds_dwordx4): one instruction moves
// 1 KiB = the 256 B (16 pieces of 16 B: one ARTM call, four PCM/FM calls) of four consecutive chunks, so HBM sees
// whole 128 B lines, never a lane-strided gather.  The ring image is lane-linear (the DMA's destination is
// M0 + lane * 16), so the bank swizzle is applied on the SOURCE side: slot j of chunk c holds piece j ^ (c & 15), and
// lane c reads piece q at slot q ^ (c & 15) — conflict-free ds_read_b128 (the 16 lanes of a service group hit 16
// different 4-bank groups).  The DMAs are inline asm (hipcc drains a builtin LDS-DMA with vmcnt(0) before the next
// ds_read; MI355X guide, "Pipelining across barriers"): R batches in flight per wave, retired by a counted
// s_waitcnt vmcnt(16 (R - 1)); the ring is private to its wave, so no barrier is involved.
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

#define LANE_BATCH_BYTES 16384          // one ring slot: 64 chunks x 16 pieces x 16 B
#define LANE_DMAS 16                    // LDS-DMA instructions per batch (1 KiB each)
#define LANE_ROT_GAP 2056               // bytes between the cos and the sin column of the rotation table
#define LANE_ROT_BYTES (LANE_ROT_GAP + 64 * 8)

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
    static_assert((M == 2 || M == 4) && LP >= 2 && S <= 16 && NF <= 16 && 16 % NF == 0 && P % NC == 0 && 2 * P <= 64,
                  "lane form: trellis of <= 16 states, <= 16 filters per call, pulse of >= 2 symbols");
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
// written and restored inside the statement — the compiler owns M0 everywhere else).  Rows are read once: nt.
__device__ __forceinline__ void lane_dma4(unsigned v0, unsigned v1, unsigned v2, unsigned v3, const void *sbase, unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %5 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %3, %5 nt\n\t"
                 "s_add_u32 m0, m0, 0x400\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, %5 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds)
                 : "memory", "scc");
}

// One LDS-DMA instruction with a full per-lane address (the waves at either end of the burst, whose rows are clamped).
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

template <int N>
__device__ __forceinline__ void lane_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct cpm_lane_params {
    int64_t ncalls, nchunks;
    int CH, W, D;
};

// DHI: the emitted field (bits lgM (D - 1) ... of the best state's decision register) sits in the register's high half.
template <class SP, int R, bool DHI>
__global__ __launch_bounds__(64, 1) void cpm_lane_kernel(const double2 *__restrict__ rows, const double2 *__restrict__ rot_cs,
                                                         uint8_t *__restrict__ out, uint64_t *__restrict__ state,
                                                         uint64_t *__restrict__ edge, cpm_lane_params P)
{
    constexpr int S = SP::S, M = SP::M, NF = SP::NF, LGM = SP::LGM, NC = SP::NC, TWO_P = 2 * SP::P;
    constexpr int B = 16 / NF;                                      // calls per batch
    static_assert(SP::NH == 1 || (R * B) % 2 == 0, "the leaving symbol's parity must be a constant of the unrolled step");
    static_assert(LANE_DMAS * (R - 1) <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    // Rotation table as two 8-byte columns read by two ds_read_b64 (64-bank mode: the 2p <= 64 entries of a column sit in
    // distinct banks, equal entries are broadcast).  One ds_read2st64_b64 — what the compiler makes of columns 1 KB
    // apart — is served in 32-bank mode, where entries 16 apart collide: 40 % of this kernel's LDS cycles were bank
    // conflicts (profiles/r04_pmc_lane_*).  LANE_ROT_GAP is neither a ds_read2_b64 nor a ds_read2st64_b64 distance.
    double *rot_cos = reinterpret_cast<double *>(smem + R * LANE_BATCH_BYTES);
    double *rot_sin = reinterpret_cast<double *>(smem + R * LANE_BATCH_BYTES + LANE_ROT_GAP);
    for (int k = lane; k < TWO_P; k += 64) {
        const double2 e = rot_cs[k];
        rot_cos[k] = e.x;
        rot_sin[k] = e.y;
    }
    if (blockIdx.x == 0 && lane == 0) edge[P.nchunks * CPM_EDGE_WORDS] = 0;     // cpm_verify_kernel's list of failed chunks: none yet
    const int64_t n0 = state ? lane_uniform64((int64_t)state[CPM_ST_N]) : 0;    // calls made before this launch
    const int64_t chunk0 = (int64_t)blockIdx.x * 64;
    const int64_t chunk = chunk0 + lane;
    const int64_t k_first = chunk * P.CH;
    const bool live = k_first < P.ncalls;
    // The step is unrolled with the leaving symbol's parity as a constant: the wave starts its warm-up one call
    // earlier when that makes the parity of its first call 0 (chunk starts and chunk lengths are even).
    const int Weff = P.W + (SP::NH == 2 ? (int)((n0 - P.W - SP::LP + 1) & 1) : 0);
    const int T = Weff + P.CH;                                       // calls a lane runs
    const int nb = (T + B - 1) / B;                                  // batches
    const int dshift = LGM * (P.D - 1) - (DHI ? 32 : 0);            // inside its 32-bit half

    // detector registers of this chunk
    double m[S];
    int r[S];
    uint64_t h[S];
    {
        const int64_t k_start = chunk == 0 ? 0 : k_first - Weff;     // first call this lane really runs
        const int tilt = cpm_tilt(M, SP::P, SP::NH, SP::K0, SP::K1, SP::LP, n0 + k_start);
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            m[s] = 0.0;
            const int v = 2 * (s % NC) - tilt;
            r[s] = v < 0 ? v + TWO_P : v;
            h[s] = 0;
        });
    }
    if (state && chunk == 0 && n0 > 0) {                            // continue the carried detector
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            m[s] = __longlong_as_double((long long)state[CPM_ST_M + s]);
            r[s] = (int)state[CPM_ST_V + s];
            h[s] = state[CPM_ST_H + s];
        });
    }
    uint64_t *const erec = edge + chunk * CPM_EDGE_WORDS;            // (written only when the chunk is live)
    uint64_t acc = 0;                                                // decisions of the current group of 8 calls, one byte each

    // One detector call.  KV: the leaving symbol's variant (compile-time); zsrc(f): filter output f of this call.
    auto step = [&](auto kvc, auto &&zsrc, bool emit, bool emit_ok, int group_pos) __attribute__((always_inline)) {
        constexpr int KV = decltype(kvc)::value;
        constexpr int K_old = KV == 2 ? 0 : (KV == 1 ? SP::K1 : SP::K0);
        double nm[S];
        int nr[S];
        uint64_t nh[S];
        // Start states grouped by `corr` (their Lp - 1 previous symbols): a group needs M filter outputs and NC rotations
        // and hands exactly one candidate to each of NC * M end states — in the order of the sequential statement's branch list (start
        // state ascending), so the running strict '<' below IS its first arg-min: the first listed branch keeps a tie.
        // (Grouped by end state instead, all S rotations and all M^Lp filter outputs stay live across the call: 354
        // registers for ARTM, a third of them parked in AGPRs, and as many compare masks as the SGPR file holds.)
        // A lane is alone on its SIMD for most of a burst (1e7 calls make ~600 waves for 1024 SIMDs), so nothing but
        // its own instruction stream hides an LDS round trip: the operands of group g + 1 are requested BEFORE group g
        // is worked (profiles/r04_pmc_lane_*: 29 % of the wave's cycles sat in s_waitcnt lgkmcnt without it).
        struct operands {
            double2 z[M];
            double cr[NC], sr[NC];
        };
        auto load_group = [&](auto cc, operands &o) __attribute__((always_inline)) {
            constexpr int corr = decltype(cc)::value;
            lane_for<0, M>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                o.z[u] = zsrc(std::integral_constant<int, u + M * corr>{});
            });
            lane_for<0, NC>([&](auto lc) {
                constexpr int src = decltype(lc)::value + NC * corr;
                o.cr[decltype(lc)::value] = rot_cos[r[src]];
                o.sr[decltype(lc)::value] = rot_sin[r[src]];
            });
        };
        auto run_group = [&](auto self, auto cc, const operands &o) __attribute__((always_inline)) -> void {
            constexpr int corr = decltype(cc)::value;
            constexpr int u_old = corr / SP::MSUB;                   // the symbol leaving the window: slot index of these candidates
            constexpr int inc = (K_old * u_old) % SP::P;
            constexpr int delta = ((2 * inc - (M - 1) * K_old) % TWO_P + TWO_P) % TWO_P;   // what the branch adds to the TILTED phase index
            operands nx;
            if constexpr (corr + 1 < SP::NCORR) load_group(std::integral_constant<int, corr + 1>{}, nx);
            lane_for<0, NC>([&](auto lc) {
                constexpr int cls = decltype(lc)::value;
                constexpr int src = cls + NC * corr;
                const double cr = o.cr[cls], sr = o.sr[cls];
                const uint32_t x = (uint32_t)(r[src] + delta);                              // < 4p
                const int rs = (int)min(x, x - (uint32_t)TWO_P);                          // mod 2p: the difference wraps to a huge value when x < 2p
                lane_for<0, M>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    constexpr int e = (cls + inc) % NC + NC * (u + M * (corr % SP::MSUB));
                    const double inc_m = -fma(cr, o.z[u].x, sr * o.z[u].y);              // -Re(e^{-j theta} Z)
                    const double c = m[src] + inc_m;
                    if constexpr (u_old == 0) {
                        nm[e] = c;
                        nr[e] = rs;
                        nh[e] = h[src];
                    } else {
                        const bool f = c < nm[e];
                        nm[e] = lane_min(nm[e], c);
                        nr[e] = f ? rs : nr[e];
                        nh[e] = f ? h[src] : nh[e];
                        // (pins the phase select next to its compare: nothing needs nr before the end of the call, so the
                        // optimiser sank all 48 of them — and their 48 lane masks, the whole scalar file — below the emission)
                        asm volatile("" : "+v"(nr[e]));
                    }
                });
                // (the M candidates of a start state end here for the instruction scheduler: left free, it lines up all
                // of a call's compares first and keeps their lane masks — two scalar registers each — alive until the
                // selects: 100+ scalar spills, reloaded by v_readlane inside the call loop)
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (corr + 1 < SP::NCORR) self(self, std::integral_constant<int, corr + 1>{}, nx);
        };
        {
            operands o0;
            load_group(std::integral_constant<int, 0>{}, o0);
            run_group(run_group, std::integral_constant<int, 0>{}, o0);
        }
        const double gmin = lane_tree_min<0, S, S>(nm);
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            m[s] = nm[s] - gmin;                                     // the minimum becomes exactly 0.0
            r[s] = nr[s];
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

    // Virtual pre-start symbols: the first LP - 1 calls of a fresh burst have no symbol leaving the window (variant 2).
    // Only chunk 0 meets them; it runs them here, straight from global memory, and skips them in the loop.
    int64_t kmin = 0;
    if (chunk == 0 && n0 < SP::LP - 1) {
        kmin = SP::LP - 1 - n0;
        if (kmin > P.ncalls) kmin = P.ncalls;
        for (int64_t k = 0; k < kmin; ++k) {
            const double2 *zr = rows + k * NF;
            step(kv2{}, [&](auto fc) { return zr[decltype(fc)::value]; }, true, n0 + k >= P.D - 1, (int)(k & 7));
        }
    }
    // Call t of the loop is local call kbase + t.  What a lane may run, as 32-bit bounds on t (64-bit compares per call
    // cost the loop its scalar registers): [t_lo, t_hi) = inside the burst, live, not run above; t_ok: first call that decides.
    const int64_t kbase = k_first - Weff;
    auto clamp_t = [&](int64_t v) __attribute__((always_inline)) { return (int)(v < 0 ? 0 : (v > T ? T : v)); };
    const int t_lo = clamp_t(kmin - kbase);
    const int t_hi = live ? clamp_t(P.ncalls - kbase) : 0;
    const int t_ok = clamp_t((int64_t)P.D - 1 - n0 - kbase);
    uint8_t *const outp = out + kbase;                               // out[k] = outp[t]

    // ---- the ring
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int c0 = lane >> 4, j0 = lane & 15;
    // (uniform) every row this wave will ever ask for exists: the DMAs take base + per-lane offset, no clamping
    const bool interior = blockIdx.x > 0 && (chunk0 + 64) * P.CH + B <= P.ncalls;
    unsigned voff[LANE_DMAS];
    lane_for<0, LANE_DMAS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int c = 4 * i + c0;                                    // chunk of the wave this lane fetches for in DMA i
        const int q = j0 ^ (c & 15);                                 // ... and which of its 16 pieces
        voff[i] = ((unsigned)c * (unsigned)P.CH * NF + (unsigned)q) * 16u;
    });
    auto fetch = [&](int b, int slot) __attribute__((always_inline)) {
        const int bb = b < nb ? b : nb - 1;                          // (past the end: the last batch again — the count of DMAs in flight must not change)
        const unsigned lds = lds0 + (unsigned)slot * LANE_BATCH_BYTES;
        if (interior) {
            const double2 *sb = rows + ((chunk0 * P.CH - Weff + (int64_t)bb * B) * NF);
            lane_dma4(voff[0], voff[1], voff[2], voff[3], sb, lds);
            lane_dma4(voff[4], voff[5], voff[6], voff[7], sb, lds + 4096);
            lane_dma4(voff[8], voff[9], voff[10], voff[11], sb, lds + 8192);
            lane_dma4(voff[12], voff[13], voff[14], voff[15], sb, lds + 12288);
        } else {
            // (cold: two waves of a burst.  Everything is re-derived from values made opaque here, so that none of it is
            // hoisted out of the call loop into registers the interior waves would pay for)
            int cz = c0;
            unsigned lz = lds;
            asm volatile("" : "+v"(cz), "+s"(lz));
            lane_for<0, LANE_DMAS>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int c = 4 * i + cz;
                const int q = j0 ^ (c & 15);
                int64_t row = (chunk0 + c) * P.CH - Weff + (int64_t)bb * B + q / NF;
                row = row < 0 ? 0 : (row >= P.ncalls ? P.ncalls - 1 : row);          // never decoded when clamped
                lane_dma1(rows + row * NF + (q % NF), lz + 1024u * i);
            });
        }
    };
    const unsigned lds_lane = (unsigned)lane * 256u + (unsigned)j0 * 16u;           // this chunk's 256 B of a slot, swizzle folded in

    for (int b = 0; b < R - 1; ++b) fetch(b, b);
    for (int b0 = 0; b0 < nb; b0 += R) {
        lane_for<0, R>([&](auto rc) {
            constexpr int RS = decltype(rc)::value;
            const int b = b0 + RS;
            if (b < nb) {
                fetch(b + R - 1, (RS + R - 1) % R);                  // into the slot batch b - 1 was read from
                lane_wait_vm<LANE_DMAS * (R - 1)>();                 // batch b has landed (this wave's own DMAs: no barrier)
                lane_for<0, B>([&](auto tc) {
                    constexpr int TT = decltype(tc)::value;
                    constexpr int KV = SP::NH == 2 ? ((RS * B + TT) & 1) : 0;
                    const int t = b * B + TT;
                    if (t == Weff && live) {                         // the next call is the chunk's first own one
                        lane_for<0, S>([&](auto sc) {
                            constexpr int s = decltype(sc)::value;
                            erec[3 * s] = (uint64_t)__double_as_longlong(m[s]);
                            erec[3 * s + 1] = (uint64_t)(int64_t)r[s];
                            erec[3 * s + 2] = h[s];
                        });
                    }
                    const bool emit = t >= Weff;
                    const int gp = (t - Weff) & 7;
                    if (t >= t_lo && t < t_hi) {
                        const char *zb = smem + RS * LANE_BATCH_BYTES;
                        step(std::integral_constant<int, KV>{},
                             [&](auto fc) {
                                 constexpr int q = TT * NF + decltype(fc)::value;
                                 typedef double v2d __attribute__((ext_vector_type(2)));
                                 const v2d v = *reinterpret_cast<const v2d *>(zb + (lds_lane ^ (unsigned)(q * 16)));
                                 return make_double2(v.x, v.y);
                             },
                             emit, t >= t_ok, gp);
                    }
                    if (emit && gp == 7 && t < T) {                  // a group of 8 decisions is complete (uniform)
                        if (t < t_hi) {
                            *reinterpret_cast<uint64_t *>(outp + (t - 7)) = acc;
                        } else if (t - 7 < t_hi) {                   // the burst ends inside the group
                            for (int q = 0; t - 7 + q < t_hi; ++q) outp[t - 7 + q] = (uint8_t)(acc >> (8 * q));
                        }
                        acc = 0;
                    }
                });
            }
        });
    }
    lane_wait_vm<0>();
    // proof record: what this chunk ended with
    if (live) {
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            erec[48 + 3 * s] = (uint64_t)__double_as_longlong(m[s]);
            erec[48 + 3 * s + 1] = (uint64_t)(int64_t)r[s];
            erec[48 + 3 * s + 2] = h[s];
        });
    }
    if (state && live && k_first + P.CH >= P.ncalls) {              // the lane that owns the last call
        state[CPM_ST_STAGE + CPM_ST_N] = (uint64_t)(n0 + P.ncalls);
        lane_for<0, S>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            state[CPM_ST_STAGE + CPM_ST_M + s] = (uint64_t)__double_as_longlong(m[s]);
            state[CPM_ST_STAGE + CPM_ST_V + s] = (uint64_t)(int64_t)r[s];
            state[CPM_ST_STAGE + CPM_ST_H + s] = h[s];
        });
    }
}

// ---- the compiled specialisations: the two waveforms BASELINE configs[2] and SURVEY 8(f3) name
using lane_artm16 = lane_spec<4, 2, 4, 16, 2, 4, 5>;     // ARTM multi-h CPM, h = {4/16, 5/16}, pulse truncated to 2 symbols, 4 phase classes
using lane_pcmfm10 = lane_spec<2, 2, 5, 10, 1, 7, 7>;    // PCM/FM, h = 7/10, 5 phase classes
// Ring depth R (batches of 16 KB in flight per wave) decides how many waves a CU holds (160 KB of LDS) against how far
// ahead of the detector the rows are fetched.  WF_CPM_LANE_R selects another compiled depth (tuning aid).
#define LANE_R_ARTM 4
#define LANE_R_PCMFM 3

static int lane_ring_depth(int spec)
{
    int r = spec == 0 ? LANE_R_ARTM : LANE_R_PCMFM;
    if (const char *e = getenv("WF_CPM_LANE_R")) {
        const int v = atoi(e);
        if (v == 2 || (spec == 0 && v == 4) || (spec == 1 && v == 3)) r = v;
    }
    return r;
}

int wf_cpm_lanes_plan(const wf_cpm_detector_config *d, cpm_lane_plan *plan)
{
    if (const char *e = getenv("WF_CPM_LANES"))
        if (atoi(e) == 0) return 1;
    int spec = -1;
    if (d->M == 4 && d->Lp == 2 && d->NC == 4 && d->p == 16 && d->nh == 2 && d->K[0] == 4 && d->K[1] == 5 && d->D >= 17 && d->D <= 32) spec = 0;
    if (d->M == 2 && d->Lp == 2 && d->NC == 5 && d->p == 10 && d->nh == 1 && d->K[0] == 7 && d->D >= 1 && d->D <= 32) spec = 1;
    if (spec < 0) return 1;
    const int R = lane_ring_depth(spec);
    int per_cu = (160 * 1024) / (R * LANE_BATCH_BYTES + LANE_ROT_BYTES);
    if (per_cu > 8) per_cu = 8;
    *plan = {spec, R, per_cu, spec == 0 ? 1 : 4};
    return 0;
}

int wf_cpm_lanes_launch(const cpm_lane_plan &plan, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri,
                        int64_t ncalls, int warmup, int chunk_calls, int64_t nchunks, uint8_t *d_decisions, void *d_state, uint64_t *d_edge,
                        void *stream)
{
    WF_REQUIRE(chunk_calls >= 64 && chunk_calls % 64 == 0 && warmup >= 0 && warmup % 2 == 0 && chunk_calls > warmup + 1,
               "wf_cpm_lanes: chunk of %d calls, warm-up %d", chunk_calls, warmup);
    WF_REQUIRE((int64_t)64 * chunk_calls * 16 * 16 < (1ll << 32), "wf_cpm_lanes: chunk of %d calls overflows the 32-bit row offsets", chunk_calls);
    cpm_lane_params P{ncalls, nchunks, chunk_calls, warmup, det->D};
    const int64_t nblocks = (nchunks + 63) / 64;
    WF_REQUIRE(nblocks < (1ll << 31), "wf_cpm_lanes: burst too long for one launch");
    const size_t lds = (size_t)plan.ring_batches * LANE_BATCH_BYTES + LANE_ROT_BYTES;
    using kern_t = void (*)(const double2 *, const double2 *, uint8_t *, uint64_t *, uint64_t *, cpm_lane_params);
    kern_t k;
    if (plan.spec == 0) k = plan.ring_batches == 2 ? static_cast<kern_t>(cpm_lane_kernel<lane_artm16, 2, true>) : static_cast<kern_t>(cpm_lane_kernel<lane_artm16, 4, true>);
    else k = plan.ring_batches == 2 ? static_cast<kern_t>(cpm_lane_kernel<lane_pcmfm10, 2, false>) : static_cast<kern_t>(cpm_lane_kernel<lane_pcmfm10, 3, false>);
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)nblocks), dim3(64), lds, wf_stream(stream), reinterpret_cast<const double2 *>(d_rows_ri),
                       reinterpret_cast<const double2 *>(d_rot_cs), d_decisions, static_cast<uint64_t *>(d_state), d_edge, P);
    WF_LAUNCH_CHECK();
    return WF_OK;
}
