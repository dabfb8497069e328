// wf_modulate.hip — fused CPM modulator: symbols -> complex baseband in ONE pass over HBM.
// cpm_modulate = upsample + pulse FIR ("same") + accumulate-mod-sps + exp(j.)
// (reference waveforms/cpm/modulate.py:57-101) without materialising the f64
// frequency-pulse array: 1 B read + 16*sps B written per symbol.
//
// Why no inter-workgroup scan is needed: the frequency-pulse train is a FIR of symbol
// impulses, so its running sum up to any sample is
//     T * S(m_full)  +  sum over the <= J+1 symbols still under the pulse of
//                       amp[m] * Gcum[...]  -  (head truncation constant K0)
// with T = sum(g), Gcum = cumsum(g), S = symbol-rate prefix sum of amp = symbol * h.
// Two tiny kernels build T*S at tile granularity (exact modular accumulation in 62-bit
// fixed point, like wf_phase.hip); the main kernel then handles every tile
// independently: symbol amplitudes of the tile in LDS, per-thread tap phases in
// registers (wf_fir.hip), in-tile fp64 scan by DPP + LDS wave totals (wf_phase.hip),
// carry, mod, table sincos (LDS), wave-transposed stores of 64 consecutive samples.
#include <type_traits>

#include "wf_common.h"

#ifndef MOD_THREADS
#define MOD_THREADS 256
#endif
#ifndef MOD_ROWS
#define MOD_ROWS 16
#endif
#define MOD_WAVES (MOD_THREADS / WF_WAVE)
#define MOD_MASK ((1ull << 62) - 1)
#define MOD_MAX_PART 40
#define MOD_SCAN_TAPS 2048   // taps staged in LDS by the carry kernel
#define MOD_SCAN_PER 16      // tiles per thread the carry kernel keeps in registers
#define MOD_MAX_NH 8         // modulation indices the one-pass modulator cycles (modulate.py:91-92 takes any N_h; the fused front ends: <= 2)

struct mod_params {
    int64_t nsym, out_len, ntiles;
    int sps, ntaps, nh, c, rs;
    int spt;      // symbols per tile = MOD_ROWS * rs / sps
    int dsh;      // D = ceil((ntaps - c) / sps): symbols not yet fully elapsed at a tile edge
    int npart;    // symbols partially elapsed at a tile edge
    int nhead;    // symbols whose pulse starts before sample 0
    double sps_d, inv_sps, phi0_turns;
    // window of a longer stream (all zero / full range for a one-shot burst): symbols[0] is
    // global symbol sym_origin (nloc of them resident), local tile 0 is global tile tile_lo,
    // out[0] is global sample out_origin, outputs are written for samples < out_hi
    int64_t sym_origin, nloc, tile_lo, out_origin, out_hi;
    int64_t q_out_tile;   // local tile whose carry is exported (-1: none)
};

// symbol amplitude at GLOBAL 0-based index m (0 outside the burst; the caller guarantees
// that every index a tile needs inside the burst is resident)
__device__ __forceinline__ double mod_amp(const int8_t *__restrict__ symbols, const double *__restrict__ hvec,
                                          const mod_params &P, int64_t m)
{
    const int64_t l = m - P.sym_origin;
    if (m < 0 || m >= P.nsym || l < 0 || l >= P.nloc) return 0.0;
    return (double)symbols[l] * hvec[P.nh == 1 ? 0 : (int)(m % P.nh)];
}

// scratch layout (doubles): [0] T, [1] K0, [2 .. 2+MOD_MAX_PART) Gpart, then ntiles tile sums P,
// then ntiles u64 carries Wq.
#define MOD_OFF_GPART 2
#define MOD_OFF_P (2 + MOD_MAX_PART)
// ... 8 spare words, then Gcum = running sum of the taps (ntaps doubles): the main kernels take their
// per-lane tap-phase tables from it, so tile carries and in-tile sums use the very same doubles
#define MOD_OFF_GCUM(ntiles) ((size_t)MOD_OFF_P + 2 * (size_t)(ntiles) + 8)

__device__ __forceinline__ double mod_pos_d(double v, double sps, double inv_sps)
{
    double k = floor(v * inv_sps);
    double m = fma(-k, sps, v);
    if (m < 0.0) m += sps;
    if (m >= sps) m -= sps;
    return m;
}

// P[j] = sum of amp[m] over m in [(j-1)*spt - D, j*spt - D) ∩ [0, nsym): one wave per tile.
__global__ __launch_bounds__(MOD_THREADS) void mod_tile_sums_kernel(const int8_t *__restrict__ symbols,
                                                                     const double *__restrict__ hvec, mod_params P,
                                                                     double *__restrict__ scratch)
{
    const int lane = threadIdx.x & 63;
    const int64_t j = (int64_t)blockIdx.x * MOD_WAVES + (threadIdx.x >> 6);
    if (j >= P.ntiles) return;
    const int64_t lo = (P.tile_lo + j - 1) * P.spt - P.dsh;
    double acc = 0.0;
    const int64_t l0 = lo - P.sym_origin;
    if (P.nh == 1 && lo >= 0 && lo + P.spt <= P.nsym && l0 >= 0 && l0 + P.spt <= P.nloc) {
        // interior tile, single modulation index (wave-uniform test): 8 symbols per lane and trip
        // from one unaligned 8 B load, no per-symbol bounds or index arithmetic
        const double h0 = hvec[0];
        const int8_t *sp = symbols + l0;
        int k = 8 * lane;
        for (; k + 8 <= P.spt; k += 8 * WF_WAVE) {
            const uint64_t w = *reinterpret_cast<const uint64_t *>(sp + k);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += (double)(int8_t)(w >> (8 * q)) * h0;
        }
        for (int q = 0; k + q < P.spt && q < 8; ++q) acc += (double)sp[k + q] * h0;   // spt not a multiple of 8
    } else if (P.nh == 2 && lo >= 0 && lo + P.spt <= P.nsym && l0 >= 0 && l0 + P.spt <= P.nloc) {
        // interior tile, two alternating modulation indices (ARTM): the same 8-symbols-per-load walk, index by symbol parity
        // (the generic walk below cost 19 us per 1e7 symbols against 7 for the single-index form)
        const double h0 = hvec[0], h1 = hvec[1];
        const int8_t *sp = symbols + l0;
        const bool odd0 = (lo & 1) != 0;                         // (k below is a multiple of 8: parity of lo + k + q = parity of lo + q)
        const double he = odd0 ? h1 : h0, ho = odd0 ? h0 : h1;
        int k = 8 * lane;
        for (; k + 8 <= P.spt; k += 8 * WF_WAVE) {
            const uint64_t w = *reinterpret_cast<const uint64_t *>(sp + k);
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += (double)(int8_t)(w >> (8 * q)) * ((q & 1) ? ho : he);
        }
        for (int q = 0; k + q < P.spt && q < 8; ++q) acc += (double)sp[k + q] * ((q & 1) ? ho : he);
    } else {
        for (int k = lane; k < P.spt; k += WF_WAVE) acc += mod_amp(symbols, hvec, P, lo + k);
    }
#pragma unroll
    for (int d = WF_WAVE / 2; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, WF_WAVE);
    if (lane == 0) scratch[MOD_OFF_P + j] = acc;
}

// Single block: Gcum-derived constants, then Wq[j] = fixed(sum_{i<=j} T*P[i] - K0) (mod sps).
// (held to 64 registers — 8 waves per SIMD: in a pipelined link this single workgroup of 16 waves has to find room on ONE CU beside
//  the previous block's detector, whose lane waves hold 240 of a SIMD's 512 registers each; at 76 registers it did not fit next to
//  them and waited ~0.4 ms for the first of them to retire — profiles/r06_timeline_multih_scan_blocked.txt)
__global__ __launch_bounds__(1024, 8) void mod_tile_scan_kernel(const int8_t *__restrict__ symbols,
                                                              const double *__restrict__ hvec,
                                                              const double *__restrict__ pulse, mod_params P,
                                                              double *__restrict__ scratch,
                                                              const uint64_t *__restrict__ q_in,
                                                              uint64_t *__restrict__ q_out)
{
    __shared__ double s_T, s_K0;
    __shared__ uint64_t s_part[1024 / WF_WAVE];
    __shared__ double s_pulse[MOD_SCAN_TAPS];
    const int t = threadIdx.x;
    // the serial pass below reads one tap per trip: from global memory that is a dependent
    // ~0.3 us load each (20 us for the 65-tap SOQPSK pulse), so the taps are staged first
    const bool staged = P.ntaps <= MOD_SCAN_TAPS;
    if (staged)
        for (int k = t; k < P.ntaps; k += 1024) s_pulse[k] = pulse[k];
    // ... and so are the amplitudes of the few symbols whose pulse starts before sample 0
    __shared__ double s_head[MOD_MAX_PART + 1];
    if (t >= 1 && t <= MOD_MAX_PART) s_head[t] = (t <= P.nhead && t <= P.nsym) ? mod_amp(symbols, hvec, P, t - 1) : 0.0;
    __syncthreads();
    if (staged) {
        // Gcum = running sum of the taps, in place in LDS: wave 0 scans 64 taps per trip (DPP) and
        // carries the total; the few entries the tile carries need are then gathered in parallel.
        // (A serial `run += pulse[k]` walk by one lane cost ~5 us of dependent LDS reads.)
        if (t < WF_WAVE) {
            double carry = 0.0;
            for (int k0 = 0; k0 < P.ntaps; k0 += WF_WAVE) {
                const int k = k0 + t;
                const double inc = wf_wave_incl_scan(k < P.ntaps ? s_pulse[k] : 0.0) + carry;
                if (k < P.ntaps) s_pulse[k] = inc;
                carry = __shfl(inc, WF_WAVE - 1, WF_WAVE);
            }
        }
        __syncthreads();
        int R = (P.c - P.ntaps) % P.sps;
        if (R < 0) R += P.sps;
        if (t < MOD_MAX_PART) {
            // idx_l = ntaps - 1 + R - (l+1)*sps  (partial symbols at a tile edge)
            const int idx = P.ntaps - 1 + R - (t + 1) * P.sps;
            scratch[MOD_OFF_GPART + t] = (t < P.npart && idx >= 0 && idx < P.ntaps) ? s_pulse[idx] : 0.0;
        }
        if (t == 0) {
            // head truncation: symbols whose pulse starts before sample 0, tap index c - 1 - mp1*sps
            double k0 = 0.0;
            for (int mp1 = 1; mp1 <= P.nhead && mp1 <= P.nsym; ++mp1) {
                const int idx = P.c - 1 - mp1 * P.sps;
                if (idx >= 0 && idx < P.ntaps)
                    k0 += (mp1 <= MOD_MAX_PART ? s_head[mp1] : mod_amp(symbols, hvec, P, mp1 - 1)) * s_pulse[idx];
            }
            const double T = s_pulse[P.ntaps - 1];
            scratch[0] = T;
            scratch[1] = k0;
            s_T = T;
            s_K0 = k0;
        }
        for (int k = t; k < P.ntaps; k += 1024) scratch[MOD_OFF_GCUM(P.ntiles) + k] = s_pulse[k];
    } else if (t == 0) {
        // very long pulses: one sequential pass over the taps in global memory
        int R = (P.c - P.ntaps) % P.sps;
        if (R < 0) R += P.sps;
        double run = 0.0, k0 = 0.0;
        for (int l = 0; l < MOD_MAX_PART; ++l) scratch[MOD_OFF_GPART + l] = 0.0;
        int num = P.ntaps - 1 + R, num_q = num / P.sps, num_r = num - num_q * P.sps;
        int hn = P.c - 1, hn_q = hn >= 0 ? hn / P.sps : 0, hn_r = hn >= 0 ? hn - hn_q * P.sps : 0;
        for (int k = 0; k < P.ntaps; ++k) {
            run += pulse[k];
            scratch[MOD_OFF_GCUM(P.ntiles) + k] = run;
            if (num > 0 && num_r == 0) {
                const int l = num_q - 1;
                if (l >= 0 && l < P.npart) scratch[MOD_OFF_GPART + l] = run;
            }
            if (hn > 0 && hn_r == 0) {
                const int64_t mp1 = hn_q;
                if (mp1 >= 1 && mp1 <= P.nhead && mp1 <= P.nsym)
                    k0 += (mp1 <= MOD_MAX_PART ? s_head[mp1] : mod_amp(symbols, hvec, P, mp1 - 1)) * run;
            }
            --num;
            if (--num_r < 0) { num_r = P.sps - 1; --num_q; }
            --hn;
            if (--hn_r < 0) { hn_r = P.sps - 1; --hn_q; }
        }
        scratch[0] = run;
        scratch[1] = k0;
        s_T = run;
        s_K0 = k0;
    }
    __syncthreads();
    const double T = s_T;
    const double *Pj = scratch + MOD_OFF_P;
    uint64_t *Wq = reinterpret_cast<uint64_t *>(scratch + MOD_OFF_P + P.ntiles);
    const int64_t per = (P.ntiles + 1023) / 1024;
    const int64_t j0 = (int64_t)t * per, j1 = min(P.ntiles, j0 + per);
    auto fixed_raw = [&](int64_t j) -> uint64_t {
        return (uint64_t)(mod_pos_d(T * Pj[j], P.sps_d, P.inv_sps) * P.inv_sps * 0x1.0p62) & MOD_MASK;
    };
    // a stream window that does not start at tile 0 takes the carry of its FIRST tile from the
    // previous window (its own P[0] would need symbols that are no longer resident)
    const bool carried = q_in != nullptr && P.tile_lo > 0;
    auto fixed = [&](int64_t j) -> uint64_t {
        if (carried && j == 0) return 0;
        return fixed_raw(j);
    };
    // Each thread owns `per` consecutive tiles.  Up to MOD_SCAN_PER of them are fetched with one
    // batch of independent loads and kept in registers for both passes; a serial
    // `run += fixed(j)` loop pays a full memory round trip per tile, twice.
    const bool batched = per <= MOD_SCAN_PER;
    uint64_t fx[MOD_SCAN_PER];
    uint64_t run = 0;
    if (batched) {
        double pv[MOD_SCAN_PER];
#pragma unroll
        for (int i = 0; i < MOD_SCAN_PER; ++i) pv[i] = j0 + i < j1 ? Pj[j0 + i] : 0.0;
#pragma unroll
        for (int i = 0; i < MOD_SCAN_PER; ++i) {
            fx[i] = 0;
            if (j0 + i < j1 && !(carried && j0 + i == 0))
                fx[i] = (uint64_t)(mod_pos_d(T * pv[i], P.sps_d, P.inv_sps) * P.inv_sps * 0x1.0p62) & MOD_MASK;
            run += fx[i];
        }
    } else {
        for (int64_t j = j0; j < j1; ++j) run += fixed(j);
    }
    // block-wide exclusive scan of the per-thread sums (u64, wraps mod 2^64 — only the low
    // 62 bits are kept): wave shuffles, then the 16 wave totals through LDS
    const int lane = t & 63, wave = t >> 6;
    uint64_t inc = run;
#pragma unroll
    for (int d = 1; d < WF_WAVE; d <<= 1) {
        const uint64_t o = (uint64_t)__shfl_up((unsigned long long)inc, d, WF_WAVE);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    uint64_t base = carried ? *q_in : ((uint64_t)(mod_pos_d(-s_K0, P.sps_d, P.inv_sps) * P.inv_sps * 0x1.0p62) & MOD_MASK);
    for (int w = 0; w < wave; ++w) base += s_part[w];
    run = base + inc - run;   // exclusive prefix of this thread
    uint64_t exported = 0;
    bool have = false;
    if (batched) {
#pragma unroll
        for (int i = 0; i < MOD_SCAN_PER; ++i) {
            const int64_t j = j0 + i;
            if (j < j1) {
                run += fx[i];
                Wq[j] = run & MOD_MASK;
                if (j == P.q_out_tile) {
                    exported = run & MOD_MASK;
                    have = true;
                }
            }
        }
    } else {
        for (int64_t j = j0; j < j1; ++j) {
            run += fixed(j);
            Wq[j] = run & MOD_MASK;
            if (j == P.q_out_tile) {
                exported = run & MOD_MASK;
                have = true;
            }
        }
    }
    // q_in and q_out may be the same word: every read of q_in happened before this barrier
    __syncthreads();
    if (have && q_out) *q_out = exported;
}


// ---- In-tile phase from the pulse's own prefix sums ("phase tree") -----------------------------
// The running sum of the frequency-pulse train at sample n is (file header)
//     cum[n] = T * S(m_new - J)  +  sum_{j < J} amp[m_new - j] * Gcum[min(r + j sps, ntaps - 1)]  -  K0,
// m_new = the newest symbol under the pulse at n, r = its tap index there, J * sps >= ntaps.  The tile
// carry Wq IS  T * S(last symbol fully elapsed at the tile edge) - K0  (62-bit fixed point), and what
// is left of the prefix, S(m_new - J) - S(edge), is h times an INTEGER: a sum of raw symbols.  So the
// window of symbol amplitudes a tile stages in LDS gets a companion: 32-bit prefix counts of the raw
// symbols (one set per modulation index), built once per tile by a block scan of ~4 symbols per
// thread.  A sample then costs J multiply-adds + one count read — no scan across the row's 512
// samples, no wave totals through LDS, no workgroup barrier, and no rounding that grows along the row
// (the first version ran FIR -> 64-lane DPP scan -> wave totals -> carry per row: 20 DPP moves, a
// barrier and ~25 % of the modulator's instructions).
//   s_amp[k] = amplitude of window symbol k  (global 0-based index m0 + k; 0 outside the burst)
//   s_pi[c * (win + 1) + k] = sum of the raw symbols of index class c (m % nh) over window symbols < k
template <class PT>
__device__ __forceinline__ int mod_sym_raw(const int8_t *__restrict__ symbols, const PT &P, int64_t m)
{
    const int64_t l = m - P.sym_origin;
    return (m < 0 || m >= P.nsym || l < 0 || l >= P.nloc) ? 0 : (int)symbols[l];
}

__device__ __forceinline__ int mod_wave_incl_scan_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// Called by all MOD_THREADS threads; leaves with a barrier (window and counts visible).  s_wtot: 2 * MOD_WAVES ints.
template <bool MANY_H = false, class PT>   // PT: mod_params, or the same struct read through the kernarg segment (constant address space); MANY_H: more than two modulation indices allowed
__device__ __forceinline__ void mod_stage_window(const int8_t *__restrict__ symbols, const double *__restrict__ hvec,
                                                 const PT &P, int64_t m0, int win, double *s_amp, int *s_pi, int *s_wtot,
                                                 int t = threadIdx.x)
{
    const int lane = t & 63, wave = t >> 6;
    const int per = (win + MOD_THREADS) / MOD_THREADS;          // ceil((win + 1) / MOD_THREADS): entry `win` is the total
    const int k0 = t * per;
    // a window inside the burst and resident (all but the first and last tiles): no per-symbol tests
    const int64_t l0 = m0 - P.sym_origin;
    const bool inner = m0 >= 0 && m0 + win <= P.nsym && l0 >= 0 && l0 + win <= P.nloc;
    const int8_t *sp = symbols + l0;
    if (MANY_H && P.nh > 2) {
        // Any number of modulation indices (modulate.py:91-92: symbol i takes h[i mod N_h]): one set of prefix counts per
        // index class, built class by class with the same block scan — only the stand-alone modulator comes here.
        const int nh = P.nh;
        int c0 = (int)(m0 % nh);
        c0 += c0 < 0 ? nh : 0;                                  // class of window symbol 0 (m0 < 0 in tile 0)
        auto sym_g = [&](int k) __attribute__((always_inline)) { return inner ? (int)sp[k] : mod_sym_raw(symbols, P, m0 + k); };
        for (int e = 0; e < per; ++e) {
            const int k = k0 + e;
            if (k < win) s_amp[k] = (double)sym_g(k) * hvec[(c0 + k) % nh];
        }
        for (int c = 0; c < nh; ++c) {
            int sa = 0;
            for (int e = 0; e < per; ++e) {
                const int k = k0 + e;
                if (k < win && (c0 + k) % nh == c) sa += sym_g(k);
            }
            const int ia = mod_wave_incl_scan_i32(sa);
            if (lane == 63) s_wtot[wave] = ia;
            wf_lds_barrier();
            int ra = ia - sa;
            for (int w = 0; w < wave; ++w) ra += s_wtot[w];
            for (int e = 0; e < per; ++e) {
                const int k = k0 + e;
                if (k <= win) {
                    s_pi[c * (win + 1) + k] = ra;
                    if (k < win && (c0 + k) % nh == c) ra += sym_g(k);
                }
            }
            wf_lds_barrier();                                   // s_wtot is free again; after the last class: window and counts visible
        }
        return;
    }
    const int par0 = P.nh > 1 ? (int)(m0 & 1) : 0;              // (nh <= 2) class of window symbol k = (par0 + k) & (nh - 1)
    const double h0 = hvec[0], h1 = P.nh > 1 ? hvec[1] : 0.0;
    auto sym_at = [&](int k) __attribute__((always_inline)) { return inner ? (int)sp[k] : mod_sym_raw(symbols, P, m0 + k); };
    int sa = 0, sb = 0;
    for (int e = 0; e < per; ++e) {
        const int k = k0 + e;
        if (k < win) {
            const int a = sym_at(k);
            const bool second = P.nh > 1 && ((par0 + k) & 1) != 0;
            s_amp[k] = (double)a * (second ? h1 : h0);
            sa += second ? 0 : a;
            sb += second ? a : 0;
        }
    }
    const int ia = mod_wave_incl_scan_i32(sa), ib = mod_wave_incl_scan_i32(sb);
    if (lane == 63) {
        s_wtot[wave] = ia;
        s_wtot[MOD_WAVES + wave] = ib;
    }
    wf_lds_barrier();
    int ra = ia - sa, rb = ib - sb;                      // exclusive prefix of this thread's first symbol
    for (int w = 0; w < wave; ++w) {
        ra += s_wtot[w];
        rb += s_wtot[MOD_WAVES + w];
    }
    for (int e = 0; e < per; ++e) {
        const int k = k0 + e;
        if (k <= win) {                                 // k == win: the total
            s_pi[k] = ra;
            if (P.nh > 1) s_pi[win + 1 + k] = rb;
            if (k < win) {
                const int a = sym_at(k);
                const bool second = P.nh > 1 && ((par0 + k) & 1) != 0;
                ra += second ? 0 : a;
                rb += second ? a : 0;
            }
        }
    }
    wf_lds_barrier();
}

// The two phases (mod sps, in pulse units) of the sample pair of one thread in one row.
//   a  = &s_amp[index of the newest symbol under the first sample, + 1]      (the old FIR's window top)
//   pi = &s_pi[index of the first symbol NOT yet elapsed for the first sample]  (= q0 - cq + row offset)
// wrap: the second sample already sees the next symbol (per lane; any_wrap: some lane of the kernel does).
// nh > 2 (stand-alone modulator only): classes 2 .. nh-1 take their reference counts from pi_ref[c pstride] (= s_pi of
// class c at the tile-edge index) and their index from hv[c]; T = sum of the taps.
template <int JMAX>
__device__ __forceinline__ void mod_pair_phase(const double (&Q0)[JMAX], const double (&Q1)[JMAX], const double *a, const int *pi,
                                               int wrap, bool any_wrap, int nh, int pstride, int ref_a, int ref_b, double W,
                                               double Th_a, double Th_b, double sps_d, double inv_sps, double &ra, double &rb,
                                               const int *pi_ref = nullptr, const double *hv = nullptr, double T = 0.0)
{
    const double *a1 = any_wrap ? a + wrap : a;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        acc0 = fma(Q0[j], a[-1 - j], acc0);
        acc1 = fma(Q1[j], a1[-1 - j], acc1);
    }
    double b0 = fma((double)(pi[0] - ref_a), Th_a, W);
    if (nh > 1) b0 = fma((double)(pi[pstride] - ref_b), Th_b, b0);
    if (nh > 2)
        for (int c = 2; c < nh; ++c) b0 = fma((double)(pi[c * pstride] - pi_ref[c * pstride]), T * hv[c], b0);
    double b1 = b0;
    if (any_wrap) {
        b1 = fma((double)(pi[wrap] - ref_a), Th_a, W);
        if (nh > 1) b1 = fma((double)(pi[pstride + wrap] - ref_b), Th_b, b1);
        if (nh > 2)
            for (int c = 2; c < nh; ++c) b1 = fma((double)(pi[c * pstride + wrap] - pi_ref[c * pstride]), T * hv[c], b1);
    }
    const double v0 = b0 + acc0, v1 = b1 + acc1;
    // one reduction mod sps per pair (the second sample is a single frequency-pulse value further); a
    // remainder that rounds to -eps or sps+eps needs no fix-up because the sector split of
    // wf_sincos_sectors is exact for any argument
    const double kq = floor(v0 * inv_sps);
    ra = fma(-kq, sps_d, v0);
    rb = fma(-kq, sps_d, v1);
}

#ifndef MOD_MIN_WAVES
#define MOD_MIN_WAVES 1
#endif
// FULLROW: the row is exactly 2 * MOD_THREADS samples (sps divides 512: every thread is active and
// every column is in the row), so the per-lane activity tests and their zero fills drop out.
// The arguments as they lie in the kernarg segment; the kernel reads them through this view from a pointer
// made opaque once per tile (see mcb_kargs below: held by value, the struct's ~35 uniform words stay live
// across the kernel and were spilled around the row loop — 21–32 SGPR spills, reloads inside the rows).
struct mod_kargs {
    const int8_t *symbols;
    const double *hvec, *pulse, *scratch;
    double *out;
    mod_params P;
};
typedef const __attribute__((address_space(4))) mod_kargs *mod_kptr;

// MANY_H: three or more modulation indices (launched for those only: its uniform class bookkeeping costs the common forms scalar spills).
template <int JMAX, bool FULLROW, bool MANY_H = false>
__global__ __launch_bounds__(MOD_THREADS, MOD_MIN_WAVES) void mod_main_kernel(const int8_t *__restrict__ symbols_,
                                                                const double *__restrict__ hvec_,
                                                                const double *__restrict__ pulse_,
                                                                const double *__restrict__ scratch_,
                                                                double *__restrict__ out_, mod_params P_)
{
    mod_kptr const KA0 = (mod_kptr)__builtin_amdgcn_kernarg_segment_ptr();
    const auto &P = KA0->P;
    const double *__restrict__ const hvec = KA0->hvec;
    const double *__restrict__ const scratch = KA0->scratch;
    extern __shared__ double s_amp[];       // window of symbol amplitudes, then its prefix counts (ints)
    __shared__ int s_wtot[2 * MOD_WAVES];
    __shared__ double2 s_xp[2 * MOD_THREADS];   // wave-private transpose: pairs per lane -> rows of 64 samples
    __shared__ double2 s_cis[128];              // sincos sector table (LDS copy: see wf_sincos_sectors)
    wf_stage_cis_table(s_cis, threadIdx.x, MOD_THREADS);   // the tile loop starts with a barrier
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int sps = P.sps;
    const bool active = FULLROW || 2 * t < P.rs;
    const int tile_len = MOD_ROWS * P.rs;
    const int sym_per_row = P.rs / sps;
    const int cq = P.c / sps;
    const int q0 = (2 * t + P.c) / sps;
    const int r0 = (2 * t + P.c) - q0 * sps;
    const int wrap = (r0 + 1 == sps) ? 1 : 0;
    const int r1 = wrap ? 0 : r0 + 1;
    const bool any_wrap = (sps & 1) != 0 || (P.c & 1) != 0;      // r0 = (2 t + c) mod sps is even otherwise
    // per-lane tap phases of the two samples: Gcum at r + j sps, clamped to the last tap (= T)
    const double *Gcum = scratch + MOD_OFF_GCUM(P.ntiles);
    double Q0[JMAX], Q1[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k0 = r0 + j * sps, k1 = r1 + j * sps;
        Q0[j] = Gcum[k0 < P.ntaps ? k0 : P.ntaps - 1];
        Q1[j] = Gcum[k1 < P.ntaps ? k1 : P.ntaps - 1];
    }
    const double sec_per_unit = 128.0 * P.inv_sps, sec_phi0 = 128.0 * P.phi0_turns;   // phase units -> 1/128 turns
    const int l_top0p1 = (q0 - cq) + JMAX;
    const int win = MOD_ROWS * sym_per_row + JMAX + 2;
    int *s_pi = reinterpret_cast<int *>(s_amp + ((win + 1) & ~1));
    const double T = scratch[0];
    const double Th_a = T * hvec[0], Th_b = P.nh > 1 ? T * hvec[1] : 0.0;
    const int lpart = JMAX - cq - P.dsh;        // window index of the first symbol not fully elapsed at the tile edge

    for (int64_t tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        auto kt = KA0;
        asm volatile("" : "+s"(kt));
        const auto &P = kt->P;
        const int8_t *__restrict__ const symbols = kt->symbols;
        const double *__restrict__ const hvec = kt->hvec;
        double *__restrict__ const out = kt->out;
        const uint64_t *Wq = reinterpret_cast<const uint64_t *>(kt->scratch + MOD_OFF_P + P.ntiles);
        int tp = t;                                       // (opaque per tile: the staging code's lane predicates and indices are not loop invariants to be spilled)
        asm volatile("" : "+v"(tp));
        const int64_t tile_g = P.tile_lo + tile;          // global tile index
        const int64_t tile_base = tile_g * tile_len;      // global sample index
        const int64_t sym_base = tile_base / sps;
        const int64_t mp1_lo = sym_base + cq - JMAX + 1;
        const bool full_tile = tile_base >= P.out_origin && tile_base + tile_len <= P.out_hi;
        wf_lds_barrier();                                 // the previous tile's rows are done with the window
        mod_stage_window<MANY_H>(symbols, hvec, P, mp1_lo - 1, win, s_amp, s_pi, s_wtot, tp);
        // carry into the tile: T * S(symbols fully elapsed at the tile edge) - K0, fixed point, from the
        // scan kernel (tile 0: -K0, the head truncation; a stream window: the previous window's export)
        const double W = (double)Wq[tile] * 0x1.0p-62 * P.sps_d;
        const int ref_a = s_pi[lpart], ref_b = P.nh > 1 ? s_pi[win + 1 + lpart] : 0;
        // Row by row: J multiply-adds per sample against the staged amplitudes + the prefix count ->
        // mod, sincos, store.  Rows are independent: no barrier inside the tile.
#pragma unroll 2
        for (int u = 0; u < MOD_ROWS; ++u) {
            double2 e0 = make_double2(0.0, 0.0), e1 = e0;   // samples 2t and 2t + 1 of the row
            if (active) {
                double ra, rb;
                mod_pair_phase<JMAX>(Q0, Q1, &s_amp[l_top0p1 + u * sym_per_row], &s_pi[(q0 - cq) + u * sym_per_row], wrap, any_wrap,
                                     MANY_H ? P.nh : (P.nh > 1 ? 2 : 1), win + 1, ref_a, ref_b, W, Th_a, Th_b, P.sps_d, P.inv_sps, ra, rb,
                                     MANY_H ? &s_pi[lpart] : nullptr, hvec, T);
                wf_sincos_sectors(s_cis, fma(ra, sec_per_unit, sec_phi0), &e0.y, &e0.x);
                wf_sincos_sectors(s_cis, fma(rb, sec_per_unit, sec_phi0), &e1.y, &e1.x);
            }
            // A lane holding two adjacent samples would store 16 B at a 32 B stride (half-filled
            // lines per instruction: 3.9 TB/s against 5.1 for contiguous rows), so the wave's 128
            // samples are transposed through its private LDS strip: each store instruction then
            // writes 64 consecutive samples = 1 KB.  Same wave, in-order LDS: no workgroup barrier.
            double2 *xw = s_xp + wave * (2 * WF_WAVE);
            xw[2 * lane] = e0;
            xw[2 * lane + 1] = e1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double2 xa = xw[lane], xb = xw[WF_WAVE + lane];
            const int col = wave * (2 * WF_WAVE) + lane;                       // sample column inside the row
            const int64_t na = tile_base + (int64_t)u * P.rs + col, nb = na + WF_WAVE;
            double2 *o = reinterpret_cast<double2 *>(out);
            if (full_tile) {   // block-uniform: no per-lane 64-bit window tests
                if (FULLROW || col < P.rs) wf_store16_nt(o + (na - P.out_origin), xa);
                if (FULLROW || col + WF_WAVE < P.rs) wf_store16_nt(o + (nb - P.out_origin), xb);
            } else {
                if ((FULLROW || col < P.rs) && na >= P.out_origin && na < P.out_hi) wf_store16_nt(o + (na - P.out_origin), xa);
                if ((FULLROW || col + WF_WAVE < P.rs) && nb >= P.out_origin && nb < P.out_hi) wf_store16_nt(o + (nb - P.out_origin), xb);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused modulator + channel + pulse-truncation bank (link `fuse` bit 3).  The body of
// mod_main_kernel<JMAX, true> (row length 512 = 64 symbols at sps 8) with its output stage
// replaced: instead of storing the 1.28 GB of clean baseband samples to HBM for the channel
// kernel to read back (2.56 GB of the 3.3 GB a link step moves), every thread applies the
// channel of wf_awgn_c128 to its two samples (one Philox block per thread and row) and parks
// them in a 2-row LDS ring laid out like mf_bank_kernel's window; the 64 detector columns whose
// 9-sample window starts in row u-1 are then taken from the ring by ALL 256 threads — a quad
// per column, one real sum chain per lane (the A, B, C, D of the pulse-truncation bank's shared
// sums; z1 = the plain sums) — and leave as the detector-packed 32 B row, 8 B per lane.
// Arithmetic and its order are those of mod_main_kernel, awgn / mf_bank_kernel<3,true,8,9,true>:
// the rows are bit-identical to fuse = 7.
//   Tile edge: the last column of a tile looks up to 8 samples into the next tile.  The first
// row of a tile is computed from the tile's own analytic carry, so wave 0 recomputes exactly
// those samples of tile + 1 (same expressions as that tile's row 0) — no hand-off between
// workgroups.  Column 0 of the burst looks at samples before sample 0: zeros (row -1 of tile 0).
struct mcb_params {
    double rot_re, rot_im, sigma;
    uint64_t seed, stream_id, pair0;     // pair0 = first_index / 2 (first_index even)
    const uint64_t *dyn_index;           // optional device addend to the noise index (graph replay of a stream; even)
    int64_t k_lo, k_hi;                  // columns [k_lo, k_hi) of the burst are stored, column k at rows[4 (k - k_lo)]
    int kshift;       // floor((first - 4) / 8): column k's window starts at sample 8 (k + kshift) + d
    int d;            // (first - 4) mod 8
    int pack_par0;    // parity of the detector call index of column 0 of the burst
    int cpm_nh;       // CPMNF instantiations: number of modulation-index columns of the templates (1 or 2)
    int mf_ntaps;     // PAM form (CPMNF = -1): taps of the 3-filter bank (odd, <= MCB_PAM_NT)
    int n_long, run_long;   // runs of tiles per workgroup: the first n_long workgroups take run_long tiles each, the rest ONE tile
};

// Geometry of the 2-row ring per samples-per-symbol SPS (8: the BASELINE configuration; 10: the reference's own
// examples/soqpsk_detection.py:38; 20: examples/pcmfm_test.py:25).  A row is the largest multiple of lcm(2, SPS)
// samples 256 threads x 2 samples cover (512 / 510 / 500); the ring is indexed by (sample - d) mod RING with one
// pad slot per SPS samples, so the SPS + 1 samples of a pulse-truncation window sit at slot offsets 0 .. SPS-1 and
// SPS + 1 of the pad group they start in.
template <int SPS>
struct mcb_geom {
    static constexpr int L2 = (SPS % 2 == 0) ? SPS : 2 * SPS;      // lcm(2, SPS)
    static constexpr int RS = (2 * MOD_THREADS / L2) * L2;           // samples per row
    static constexpr int CPR = RS / SPS;                             // symbols (= detector columns) per row
    static constexpr int NT = SPS + 1;                               // taps of the pulse-truncation bank
    static constexpr int RING = 2 * RS;                              // samples
    static constexpr int GROUPS = RING / SPS;                        // pad groups
    static constexpr int GS = SPS + 1;                               // slots per pad group
    static constexpr int SLOTS = GROUPS * GS + 16;                   // + the wrap copy of index RING (= 0 one turn later)
};
#define MCB_RING 1024                        // samples: 2 rows of 512 (SPS = 8)
#define MCB_SLOTS (MCB_RING + MCB_RING / 8 + 16)  // one pad slot per 8 samples (conflict-free column reads) + the wrap copy
static_assert(mcb_geom<8>::RING == MCB_RING && mcb_geom<8>::SLOTS == MCB_SLOTS, "SPS = 8 geometry");

template <int CTRL>
__device__ __forceinline__ double mcb_quad_bcast(double v)   // quad_perm CTRL (2 bits per lane: its source lane in the quad)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// CPMNF = 0: the SOQPSK pulse-truncation bank with detector-packed rows (above).  CPMNF = 4 | 16: the
// matched-filter rows of the generic CPM detector (wf_cpm_detect.hip: CPMNF templates of 9 taps per
// modulation-index column, Z = sum_k r[8 n + start + k] conj(T[n % nh][f][k]), full complex rows of
// CPMNF filters) — thread = (symbol, quarter of the filters), templates in LDS, same accumulation
// order as cpm_mf_rows_kernel.
// Waves per SIMD the register allocator is held to (the unified file has 512 registers per lane):
// the SOQPSK form fits 4 (<= 128 registers; left alone it takes 131 = 3 waves and runs 7 % slower),
// the CPM forms with a short pulse 3 (<= 168; 170 = 2 waves otherwise); nothing spills at these.
#ifndef WF_MCB_WAVES
#define WF_MCB_WAVES 4
#endif
#ifndef WF_MCB_CPM_WAVES
#define WF_MCB_CPM_WAVES 4    // CPM forms with a short pulse: 4 waves per SIMD (<= 128 registers, <= 40 KB LDS per workgroup)
#endif
#ifndef WF_MCB_UNROLL
#define WF_MCB_UNROLL 2
#endif
// Analysis aid, 0 in every build that ships: tools/section_budget.py compiles this file with ONE section of the row body
// stubbed at a time and attributes the difference in the row loop's instruction count to that section.  Bits: 1 Philox
// rounds, 2 Box-Muller transforms, 4 modulator (amplitude / count reads, phase sums, table sincos), 8 the bank (reads, sums,
// packed-row store).  Results of such a build are wrong by construction.
#ifndef WF_MCB_SECTION_OFF
#define WF_MCB_SECTION_OFF 0
#endif
// Issue priority (s_setprio): a row is a latency-bound part (amplitude / count reads from LDS, the phase sums, the
// ring writes, the bank's operand reads) and a throughput-bound part (Philox + Box-Muller: ~45 % of the vector
// instructions, next to no memory).  With the first at raised priority a wave's LDS round trips start as early as they
// can and the other waves' noise arithmetic fills the gaps.  Same-box A/B at 1e7 symbols (profiles/r03_ab_prio.log):
// no priorities 0.4878 / 0.4880 ms, the stretch between the two row barriers raised 0.4816 / 0.4850, everything but
// the noise raised (mode 2) 0.4813 / 0.4760, only the noise raised 0.4838 / 0.4902; on a second box mode 2 at priority
// 3 | 2 | 1 | none: 0.466 / 0.472 | 0.469 / 0.479 | 0.474 / 0.471 | 0.493 / 0.488, and with only the Philox rounds (no memory
// access at all) at low priority 0.489 / 0.481 against 0.470 / 0.473: the Box-Muller half must stay low, too.
#define WF_MCB_PRIO 3        // kept form: everything but the noise arithmetic at priority 3
// The kernel's arguments as they lie in the kernarg segment (each at its natural alignment, in order).
// The kernel reads them THROUGH this view, from a pointer it makes opaque once per tile: left as plain
// by-value arguments, the ~70 uniform words of the two structs are loaded once at the top and stay live
// across the whole kernel — with the row loop's own uniforms that was 88 SGPR spills (v_writelane /
// v_readlane into two reserved VGPRs, ~200 reloads per tile) and 16 VGPR spills in <9,0>.  Re-reading a
// field where it is used is a scalar load that hits the constant cache.
struct mcb_kargs {
    const int8_t *symbols;
    const double *hvec, *pulse, *scratch, *mf_taps;
    double *rows;
    mod_params P;
    mcb_params Q;
};
typedef const __attribute__((address_space(4))) mcb_kargs *mcb_kptr;

// Geometry of the PAM (long-bank) form per samples-per-symbol: 8 (BASELINE) and 10 (the reference's own
// examples/soqpsk_detection.py:38 with its better detector, :158-173).  An operand row of the matrix tile = FOUR consecutive
// columns = 3 SPS + NT samples (97 / 121), one pad slot per PG = 4 SPS samples, operand rows PG samples apart: 33 / 41 slots =
// 132 / 164 words, i.e. 4 / 36 banks (mod 64) from one row to the next — the 16 lanes a ds_read_b64 serves together, one
// per operand row, all land on different banks either way.  A row of 512 samples is 16 operand rows exactly; a row of 510 is
// 12.75: there the ODD rows of the ring start their operand rows QSH = 3 columns early (ring index 480 = 12 PG: on the pad
// grid again, and an odd shift, so that column 4 i + s - 3 of an odd row — rows hold 51 columns: the parity of a column
// flips with the row — has the parity of s like column 4 i + s of an even row: ONE set of B operands serves both).  13 / 14
// operand rows per row of 51 columns; the 2 - 5 columns they compute outside the row are not stored.
template <int SPS, int SH = 4>
struct mcb_pam_geom {
    static constexpr int NT = 9 * SPS + 1;                           // longest bank: 73 / 91 taps (rho_0 of SOQPSK-TG)
    static constexpr int PG = SH * SPS;                              // samples per pad group = per operand row
    static constexpr int OPLEN = (SH - 1) * SPS + NT;                // samples an operand row spans
    static constexpr int NK = SH == 4 ? (2 * OPLEN + 3) / 4 : (OPLEN + 3) / 4;   // k-steps (4 components each): 49 / 61; factored form (one plane per operand row): 33 / 41
    static constexpr int NKW = NK / 4;                               // k-steps per wave (wave 0 takes the odd one out)
    static constexpr int QSH = (mcb_geom<SPS>::CPR & 1) ? 3 : 0;     // columns an odd row's operand rows start early
    static constexpr int OPROWS = (mcb_geom<SPS>::CPR + QSH + SH - 1) / SH;   // operand rows in use (of 16; factored form: of 8 per plane)
    static constexpr int XLEN = NT + 1;                              // samples of the next tile the last columns of a tile look at: NT - SPS + d, d < SPS (even count)
    // an odd row's last operand row starts at RS - QSH SPS + PG (OPROWS - 1) and its last (half-filled) k-step reads sample OPLEN
    // (factored form: its last k-step reads samples up to 4 NK - 1):
    static constexpr int LASTREAD = SH == 4 ? OPLEN + 1 : 4 * NK;
    static constexpr int MIRROR = (PG * (OPROWS - 1) + LASTREAD - QSH * SPS - mcb_geom<SPS>::RS + 7) & ~7;   // ring indices 0 .. MIRROR-1 once more behind the ring (72 / 104)
    static constexpr int EXT = mcb_geom<SPS>::RING + MIRROR;                                         // extended ring: indices 0 .. EXT-1
    static constexpr int SLOTS = EXT + EXT / PG + 1;
    static_assert(NK % 4 == 1, "wave 0 takes exactly one k-step more than the others");
    static_assert((mcb_geom<SPS>::RS - QSH * SPS) % PG == 0 && PG % (SH == 4 ? 8 : 16) == 0, "operand rows start on the pad grid in both rows of the ring");
    static_assert((SH == 4 ? (((mcb_geom<SPS>::CPR - QSH) & 1) == 0 && OPROWS <= 16) : OPROWS <= 8), "one B operand set for both row parities");
};
static_assert(mcb_pam_geom<8>::NT == 73 && mcb_pam_geom<8>::NK == 49 && mcb_pam_geom<8>::XLEN == 74 && mcb_pam_geom<8>::OPROWS == 16 && mcb_pam_geom<8>::MIRROR == 72, "SPS = 8 PAM geometry");
static_assert(mcb_pam_geom<10>::NT == 91 && mcb_pam_geom<10>::NK == 61 && mcb_pam_geom<10>::OPROWS == 14 && mcb_pam_geom<10>::MIRROR == 104, "SPS = 10 PAM geometry");
static_assert(mcb_pam_geom<8, 8>::NK == 33 && mcb_pam_geom<8, 8>::OPROWS == 8 && mcb_pam_geom<8, 8>::MIRROR == 72 && mcb_pam_geom<10, 8>::NK == 41 && mcb_pam_geom<10, 8>::OPROWS == 7 && mcb_pam_geom<10, 8>::MIRROR == 104, "factored PAM geometry");
#define MCB_PAM_NT(sps) (9 * (sps) + 1)
static inline int mcb_pam_slots(int sps, bool factored)
{
    return factored ? (sps == 8 ? mcb_pam_geom<8, 8>::SLOTS : mcb_pam_geom<10, 8>::SLOTS) : (sps == 8 ? mcb_pam_geom<8>::SLOTS : mcb_pam_geom<10>::SLOTS);
}
template <int JMAX, int CPMNF, int SPS = 8>     // SPS != 8: CPMNF <= 0 only (the CPM detector's 9-tap templates are an sps-8 design); CPMNF < 0: SPS 8 / 10
__global__ __launch_bounds__(MOD_THREADS, CPMNF == 0 ? WF_MCB_WAVES : (CPMNF < 0 ? 3 : (JMAX <= 4 ? WF_MCB_CPM_WAVES : 2))) void mod_chan_bank_kernel(const int8_t *__restrict__ symbols_,
                                                                     const double *__restrict__ hvec_,
                                                                     const double *__restrict__ pulse_,
                                                                     const double *__restrict__ scratch_,
                                                                     const double *__restrict__ mf_taps_,
                                                                     double *__restrict__ rows_, mod_params P_, mcb_params Q_)
{
    mcb_kptr const KA0 = (mcb_kptr)__builtin_amdgcn_kernarg_segment_ptr();
#define MCB_FRESH(ka) asm volatile("" : "+s"(ka))
    auto ka = KA0;
    const auto &P = ka->P;
    const auto &Q = ka->Q;
    const double *__restrict__ const hvec = ka->hvec;
    const double *__restrict__ const scratch = ka->scratch;
    const double *__restrict__ const mf_taps = ka->mf_taps;
    static_assert(SPS == 8 || CPMNF == 0 || (CPMNF < 0 && SPS == 10), "the CPM detector rows are an sps-8 design, the long-bank form one for 8 and 10");
    static_assert(CPMNF >= -2 && (CPMNF <= 0 || CPMNF == 4 || CPMNF == 16 || CPMNF == 32 || CPMNF == 8),
                  "CPMNF: -1 long bank of 3 complex filters, -2 the same bank given as two real filters + a 3 x 2 complex combination; 4 | 16 CPM templates, 8 | 32 = 4 | 16 conjugate-paired");
    constexpr bool PAM = CPMNF < 0;
    constexpr bool PAM2 = CPMNF == -2;                 // the factored long bank (below)
    using G = mcb_geom<SPS>;
    using GP = mcb_pam_geom<PAM ? SPS : 8, PAM2 ? 8 : 4>;
    constexpr int NT = G::NT, RS = G::RS, CPR = G::CPR;
    constexpr bool FULLROW = RS == 2 * MOD_THREADS;                              // every thread owns two samples of a row
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    // (PAM form: "row 16" runs 74 samples = 10 symbols into the next tile, not one)
    const int win = MOD_ROWS * CPR + JMAX + 2 + (CPMNF < 0 ? 10 : 0);
    double *s_amp = s_dyn;                                                       // window of symbol amplitudes
    double2 *s_ring = reinterpret_cast<double2 *>(s_dyn + ((win + 1) & ~1));     // noisy samples, 2 rows
    // (PAM form: one pad slot per 4 SPS samples over the ring AND its mirror, slot(e) = e + e / PG: mcb_pam_geom; with the
    //  pulse-truncation layout's pad per 8 samples, 144 words, every fourth operand row shared its banks, with a pad per 16 every eighth)
    constexpr int PG = PAM ? GP::PG : SPS;                                       // samples per pad group
    constexpr int RSLOTS = PAM ? GP::SLOTS : G::SLOTS;
    int *s_pi = reinterpret_cast<int *>(s_ring + RSLOTS);                        // prefix counts of the window's raw symbols
    __shared__ int s_wtot[2 * MOD_WAVES];
    __shared__ double2 s_tab[256];      // [0,128): log table, [128,256): sincos sectors (= kWfCisTab)
    constexpr bool PAIRED = CPMNF == 32 || CPMNF == 8;     // 16 (4) filters whose templates pair off as conjugates, f <-> NF - 1 - f (below)
    constexpr int NF = PAIRED ? CPMNF / 2 : CPMNF;         // filters per row
    constexpr bool USE_MFMA = NF == 16;
    // (matrix-core form: the templates are only needed to build the B operands once, before the first row —
    //  they are staged in the ring's slots instead of 4.6 KB of their own: 4 workgroups per CU instead of 3)
    __shared__ double2 s_taps_own[USE_MFMA || PAM ? 1 : (CPMNF ? 2 * NF * 9 : 3 * NT)];
    double2 *const s_taps = USE_MFMA ? s_ring : s_taps_own;
    const int t = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(t) >> 6;      // the same number as a scalar
    wf_stage_tables<1, 0>(s_tab, t, MOD_THREADS);
    if (PAM) {
        // (the B operands are built from global memory, below)
    } else if (CPMNF) {
        for (int k = t; k < Q.cpm_nh * NF * 9; k += MOD_THREADS) s_taps[k] = reinterpret_cast<const double2 *>(mf_taps)[k];
    } else if (t < 3 * NT) {
        s_taps[t] = reinterpret_cast<const double2 *>(mf_taps)[t];
    }
    const double2 *s_cis = s_tab + 128;
    const wf_tabs_lds<1, 0> tb{s_tab};
    // pulse-truncation structure of the bank (block-uniform): filter 1 all ones, filter 2 = conj(filter 0).
    // Read per lane through a non-uniform index: as scalar loads the 27 complex taps were 108 SGPRs in flight
    // at once — the kernel's register-pressure peak, before the first tile.
    bool sym_taps = CPMNF == 0;
    if (CPMNF == 0) {
        int lane_zero = 0;
        asm volatile("" : "+v"(lane_zero));
        const double2 *tg = reinterpret_cast<const double2 *>(mf_taps) + lane_zero;
#pragma unroll 1
        for (int j = 0; j < NT; ++j) {
            const double2 t0 = tg[j], t1 = tg[NT + j], t2 = tg[2 * NT + j];
            sym_taps = sym_taps && t1.x == 1.0 && t1.y == 0.0 && t2.x == t0.x && t2.y == -t0.y;
        }
    }
    const int sym_taps_i = __builtin_amdgcn_readfirstlane(sym_taps ? 1 : 0);   // one scalar word, not a 64-bit lane mask carried through the loops
    constexpr int sps = SPS;
    constexpr int sym_per_row = CPR;
    constexpr int tile_len = MOD_ROWS * RS;
    const bool active = FULLROW || 2 * t < RS;                // (SPS 10 / 20: rows of 510 / 500 samples)
    const int cq = P.c / sps;
    const int q0 = (2 * t + P.c) / sps;
    const int r0 = (2 * t + P.c) - q0 * sps;
    const int wrap = (r0 + 1 == sps) ? 1 : 0;
    const int r1 = wrap ? 0 : r0 + 1;
    const bool any_wrap = (sps & 1) != 0 || (P.c & 1) != 0;   // even sps: r0 = (2 t + c) mod sps is even unless c is odd
    // per-lane tap phases of the two samples (mod_pair_phase): Gcum at r + 8 j, clamped to the last tap
    const double *Gcum = scratch + MOD_OFF_GCUM(P.ntiles);
    double Q0[JMAX], Q1[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k0 = r0 + j * sps, k1 = r1 + j * sps;
        Q0[j] = Gcum[k0 < P.ntaps ? k0 : P.ntaps - 1];
        Q1[j] = Gcum[k1 < P.ntaps ? k1 : P.ntaps - 1];
    }
    const double sec_per_unit = 128.0 * P.inv_sps, sec_phi0 = 128.0 * P.phi0_turns;
    const int l_top0p1 = (q0 - cq) + JMAX;
    const double T = scratch[0];
    const double Th_a = T * hvec[0], Th_b = P.nh > 1 ? T * hvec[1] : 0.0;
    // bank: quad per column, one sum chain per lane
    const int mq = t >> 2, mp = t & 3;
    // The four shared sums of the pulse-truncation bank,
    //   A: Re x * Re tap | B: Im x * Im tap | C: Re x * Im tap | D: Im x * Re tap,
    // leave as  va = odd ? C + D : A - B  (packed slot 2) and  vb = odd ? A + B : D - C  (slot 3).
    // Lanes 0, 1 of the quad run the two chains of vb and lanes 2, 3 those of va, with the sign
    // folded into the tap component (a chain over negated taps is exactly the negated chain, and
    // a - b == a + (-b)): ONE quad swap + add then gives vb in lanes 0, 1 and va in lanes 2, 3.
    //   even column:  -C | D | A | -B        odd column:  A | B | C | D      (lane 0 | 1 | 2 | 3)
    // Even lanes read Re x, odd lanes Im x, so the plain sums U (z1) sit in lanes 0 / 3 as Re / Im.
    // The column parity of a lane does not change with the row when a row holds an even number of columns
    // (SPS 8: 64; tiles start on even symbols); with 51 (SPS 10) or 25 (SPS 20) it alternates row by row.
    // Tap components in LDS (as registers they cost a fifth wave per SIMD).
    __shared__ double s_tapc[CPMNF == 0 ? 2 * 4 * NT : 1];      // (SOQPSK bank only)
    if (CPMNF == 0 && t < 8 * NT) {
        const double2 tp = reinterpret_cast<const double2 *>(mf_taps)[NT - 1 - t % NT];
        const int part = (t % (4 * NT)) / NT;
        s_tapc[t] = t < 4 * NT ? ((part == 0 || part == 3) ? -tp.y : tp.x) : ((part == 0 || part == 3) ? tp.x : tp.y);
    }
    const int odd_l = (Q.pack_par0 + mq - Q.kshift) & 1;      // parity of this lane's column in an even row of an even tile-symbol base
    const int slot_l = (mp & 1) ? 4 - mp : mp;                     // packed slot this lane stores: 0, 3, 2, 1
    const double *ring_d = reinterpret_cast<const double *>(s_ring);

    // ---- CPMNF = 16 (ARTM: 16 filters of 9 taps): the bank on the MATRIX cores ------------------------------
    // Row n of the bank is a real product  [1 x 18] . [18 x 32]:  the 9 window samples as (Im r_0, Re r_0, Im r_1,
    // ...), the 32 outputs as (Re Z_0, Im Z_0, Re Z_1, ...), so 16 symbols x 32 outputs = two
    // v_mfma_f64_16x16x4_f64 tiles over K = 20 (18 + 2 zeros): 10 MFMAs per 16 symbols in place of 144 fp64 FMAs
    // per thread and row on the vector pipe — the same flops at the same peak rate (78.6 TF either way on this
    // chip; nor does it overlap the other waves' fp64 vector work to any useful degree, see the PAM form below),
    // but without the 288 LDS operand reads per thread and row those FMAs needed.  The instruction accumulates k ascending as a bitwise fma chain (tools/mfma_f64_probe.hip:
    // 1 024 000 / 1 024 000 elements equal), which is the order cpm_oracle.c states, so the rows are those of the
    // vector-pipe kernels bit for bit.  The templates alternate with the symbol parity (two modulation indices),
    // so a wave takes 16 symbols of ONE parity: wave w = (half of the row, parity): columns 32 (w >> 1) + 2 i + (w & 1).
    // Operand layout (MI355X_MICROARCH.md): A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D[i][j] in lane
    // j + 16 (i & 3), register i >> 2.
    typedef double mcb_d4 __attribute__((ext_vector_type(4)));
    // CPMNF = 32: the same 16 filters when their templates pair off as CONJUGATES, T[15 - f] == conj(T[f]) — the symmetric
    // alphabet of cpm_detect.py:cpm_templates (alpha -> -alpha negates the phase; the caller vouches: wf_cpm_link_config.fuse
    // bit 6).  With r = x + j y and T_p = c + j s (p = 0 .. 7):  Z_p = (P + Q) + j (R - S),  Z_{15-p} = (P - Q) + j (R + S),
    // P = sum x c, Q = sum y s, R = sum y c, S = sum x s — four REAL 9-tap sums per pair, each over ONE plane of the samples.  Two
    // tiles over K = 12 (9 taps + 3 zeros): X = Re plane against (c_0 .. c_7 | -s_0 .. -s_7), Y = Im plane against
    // (s_0 .. s_7 | c_0 .. c_7): 6 matrix instructions per 16 symbols instead of 10 (each 64 cycles of its SIMD); lane j then
    // holds u = X, v = Y of pair j & 7 and forms u + v (Re Z_p | Im Z_p) and +-(u - v) (Re | Im Z_{15-p}).  Rows differ from the
    // k-ascending chain of the 10-instruction form (and of cpm_oracle.c) in the last bits: four partial sums added in another order.
    double bmat[USE_MFMA ? (PAIRED ? 6 : 10) : 1];           // B[k-step kk][output block nb], this lane's element
    const int mf_i = t & 15, mf_kq = (t & 63) >> 4;          // MFMA row / k index of this lane (A), = column / k index (B)
    if constexpr (PAIRED && USE_MFMA) {
        wf_lds_barrier();                                    // templates staged above
        const int par = (wave_u & 1) ^ (Q.kshift & 1);
        const int colT = Q.cpm_nh == 2 ? par : 0;
        const int pj = mf_i & 7;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int tap = 4 * kk + mf_kq;
            double bx = 0.0, by = 0.0;
            if (tap < 9) {
                const double2 tp = s_taps[(colT * NF + pj) * 9 + tap];
                bx = mf_i < 8 ? tp.x : -tp.y;
                by = mf_i < 8 ? tp.y : tp.x;
            }
            bmat[kk] = bx;
            bmat[3 + kk] = by;
        }
        wf_lds_barrier();                                    // every lane has its operands: the ring's slots are free again
    } else if constexpr (USE_MFMA) {
        wf_lds_barrier();                                    // templates staged above
        const int par = (wave_u & 1) ^ (Q.kshift & 1);       // modulation-index column of this wave's symbols (tiles start on even symbols)
        const int colT = Q.cpm_nh == 2 ? par : 0;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) {
                const int c = 4 * kk + mf_kq, tap = c >> 1, o = 16 * nb + mf_i, f = o >> 1;
                double v = 0.0;
                if (c < 18) {
                    const double2 tp = s_taps[(colT * NF + f) * 9 + tap];
                    // Re Z += Im r . Im T (c even) | Re r . Re T (c odd);   Im Z += Im r . Re T | Re r . (-Im T)
                    v = (o & 1) == 0 ? ((c & 1) ? tp.x : tp.y) : ((c & 1) ? -tp.y : tp.x);
                }
                bmat[nb * 5 + kk] = v;
            }
        wf_lds_barrier();                                    // every lane has its operands: the ring's slots are free again
    }

    // ---- CPMNF = -1 (PAM form: 3 filters of up to 73 complex taps, detector-packed rows) — also on the matrix cores.
    // A packed row needs 4 of a column's 6 real outputs, so FOUR consecutive columns share one operand row: row i of
    // A = the 97 samples from the window start of column 4 i (as Im, Re pairs: K = 194 -> 49 k-steps), the 16 output
    // columns of B = (shift s = 0 .. 3) x (packed slot o = 0 .. 3), B[c][4 s + o] = the coefficient of window component
    // c - 16 s in slot o of a column of parity (pack_par0 + s - kshift) & 1, zero outside the bank.  One row of 64
    // detector columns = ONE 16 x 16 tile: 49 v_mfma_f64_16x16x4_f64 (74 % of them useful flops) in place of 584 fp64
    // FMAs per column on the vector pipe.  The four waves split K (k-steps w, w + 4, ...; wave 0 takes the 49th: 13
    // operands of B per lane, registers), leave their partial tiles in LDS and each sums, in the fixed order
    // ((w0 + w1) + w2) + w3, the 16 columns it stores (register r of D = columns 16 r .. 16 r + 15: 512 B contiguous
    // per wave) — after the NEXT row barrier, which is there anyway.  Rows therefore differ from the single-chain
    // banks (mf_bank_kernel and the CPU restatement the tests check it with) in the last bits: the four chains are summed in another order.
    // (SPS 10: 121-sample operand rows, 61 k-steps, B[c][4 s + o] = the coefficient of component c - 20 s; see mcb_pam_geom)
    // ---- CPMNF = -2: the same bank FACTORED, as the reference computes it (examples/soqpsk_detection.py:158-173): two REAL
    // filters b_0, b_1 (the rho pulses, or any basis of the real row space of the three complex filters) and
    // z_s = sum_k G[s][k] (b_k * r) — mf_taps then holds b_0, b_1 (mf_ntaps doubles each) and G (3 x 2 complex).  A real filter
    // needs one PLANE of the samples per output, so an operand row is the Re or the Im parts of the 7 SPS + NT samples from the
    // window start of column 8 i: A rows 0 .. 7 = Re planes of 8 operand rows (8 columns apart), rows 8 .. 15 = their Im planes;
    // the 16 columns of B = (shift s = 0 .. 7) x (filter k), B[c][2 s + k] = tap c - SPS s of b_k.  33 (41 at sps 10) k-steps of
    // 4 samples instead of 49 (61) of 2: a third of the matrix instructions gone (each is 64 cycles of its SIMD).  D then
    // holds Re / Im of b_k * r per column; the quad that stores a column's packed row gathers its four reals (DPP) and every
    // lane forms the slot it stores: 4 multiply-adds against the 2 x 4 x 4 coefficient table s_gt (by column parity).
    __shared__ double s_gt[PAM2 ? 2 * 4 * 4 : 1];
    if constexpr (PAM2) {
        if (t < 32) {
            const int odd_c = t >> 4, o = (t >> 2) & 3, comp = t & 3, kf = comp >> 1;     // comp: Re b_0*r, Im b_0*r, Re b_1*r, Im b_1*r
            // slot 0 / 1: Re / Im z1;  slot 2: odd ? Im z0 : Re z0;  slot 3: odd ? Re z2 : Im z2
            const int sh = o < 2 ? 1 : (o == 2 ? 0 : 2);
            const bool want_im = o == 1 || (o == 2 && odd_c) || (o == 3 && !odd_c);
            const double *gp = mf_taps + 2 * Q.mf_ntaps + 2 * (2 * sh + kf);
            const double gre = gp[0], gim = gp[1];
            // Re z = sum_k Re G Re y_k - Im G Im y_k;   Im z = sum_k Im G Re y_k + Re G Im y_k
            s_gt[t] = want_im ? ((comp & 1) ? gre : gim) : ((comp & 1) ? -gim : gre);
        }
    }
    __shared__ double s_part[PAM ? 4 * 4 * 64 : 1];
    double bpam[PAM ? GP::NKW + 1 : 1];
    if constexpr (PAM2) {
        const int s_sh = mf_i >> 1, kf = mf_i & 1;
        const int nt = Q.mf_ntaps;
        const double *bf = mf_taps + kf * nt;
#pragma unroll
        for (int n = 0; n <= GP::NKW; ++n) {
            const int kk = 4 * n + wave_u;                       // this wave's k-steps: every fourth one
            const int cc = 4 * kk + mf_kq - SPS * s_sh;          // sample of this column's own window
            double v = 0.0;
            if (cc >= 0 && cc < nt && (n < GP::NKW || wave_u == 0)) v = bf[nt - 1 - cc];
            bpam[n] = v;
        }
        for (int k = t; k < RSLOTS; k += MOD_THREADS) s_ring[k] = make_double2(0.0, 0.0);
    } else if constexpr (PAM) {
        const int s_sh = mf_i >> 2, o_sl = mf_i & 3;
        const int nt = Q.mf_ntaps;
        const bool odd_col = ((Q.pack_par0 + s_sh - Q.kshift) & 1) != 0;   // (column 4 i + s of an even row = column 4 i + s - QSH of an odd one)
        const int f = o_sl < 2 ? 1 : (o_sl == 2 ? 0 : 2);
        // slot 0 / 1: Re / Im z1;  slot 2: odd ? Im z0 : Re z0;  slot 3: odd ? Re z2 : Im z2
        const bool want_im = o_sl == 1 || (o_sl == 2 && odd_col) || (o_sl == 3 && !odd_col);
        const double2 *tg = reinterpret_cast<const double2 *>(mf_taps) + f * nt;
#pragma unroll
        for (int n = 0; n <= GP::NKW; ++n) {
            const int kk = 4 * n + wave_u;                       // this wave's k-steps: every fourth one
            const int cc = 4 * kk + mf_kq - 2 * SPS * s_sh;      // component of this column's own window
            double v = 0.0;
            if (cc >= 0 && cc < 2 * nt && (n < GP::NKW || wave_u == 0)) {
                const double2 tp = tg[nt - 1 - (cc >> 1)];
                // component cc: odd = Re x, even = Im x (the imaginary sample's term first, as in every other bank)
                //   Re z += Re x . Re t + Im x . (-Im t)        Im z += Re x . Im t + Im x . Re t
                v = want_im ? ((cc & 1) ? tp.y : tp.x) : ((cc & 1) ? tp.x : -tp.y);
            }
            bpam[n] = v;
        }
        // no uninitialised slot may reach the matrix cores (0 x NaN): rows, pad slots and the mirror start as zeros
        for (int k = t; k < RSLOTS; k += MOD_THREADS) s_ring[k] = make_double2(0.0, 0.0);
    }

    // Where this thread's two samples of a row go in the ring (non-PAM forms): the ring index (sample - d) mod RING of a row
    // depends on the row's PARITY only, so the LDS byte addresses of the two slots are thread constants — both parities packed
    // into one register per sample (the static + dynamic LDS of a workgroup is < 64 KB), unpacked by one AND / one shift per
    // row where the index, pad and byte arithmetic took 13 vector instructions.  An even row also holds ring index 0 (the
    // samples before the window-start offset wrap around), whose slot is copied behind the ring for the one window that ends there.
    typedef __attribute__((address_space(3))) wf_v2d *mcb_lds2;
    constexpr bool RINGPK = CPMNF == 0 && (SPS == 8 || JMAX <= 4);     // (the CPM and PAM forms, and the long-pulse forms at 10 / 20 samples per symbol, keep the index arithmetic: two more live registers spilled there)
    unsigned ring_pk_a = 0, ring_pk_b = 0;
    // (two 16-bit LDS byte addresses per register: everything this workgroup has in LDS must end below 64 KB — the dynamic part
    //  (window, ring, prefix counts of two index classes) plus a generous 12 KB for the static tables)
    static_assert(!RINGPK || (((MOD_ROWS * CPR + JMAX + 2 + 1) & ~1) * 8 + RSLOTS * 16 + 2 * (MOD_ROWS * CPR + JMAX + 3) * 4 + 12 * 1024) < 65536,
                  "packed ring addresses need the workgroup's LDS below 64 KB");
    if constexpr (RINGPK) {
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)s_ring;
        for (int par = 0; par < 2; ++par) {
            int ia = (par ? RS : 0) + 2 * t - Q.d;
            ia += ia < 0 ? G::RING : 0;
            const int ib = ia + 1 == G::RING ? 0 : ia + 1;
            ring_pk_a |= (base + 16u * (unsigned)(ia + ia / PG)) << (16 * par);
            ring_pk_b |= (base + 16u * (unsigned)(ib + ib / PG)) << (16 * par);
        }
    }
    // A workgroup takes a RUN of consecutive tiles: rows then follow each other across the tile edge in the
    // ring exactly as inside a tile (row 15's last column completes when the next tile's row 0 is in), and
    // only the run's last tile has to compute the 8 samples after it itself (a whole extra row step for
    // wave 0: 6 % of a tile when every tile did it).
    // Long runs first, single tiles last (workgroups start in index order): the launch drains at the granularity of ONE
    // tile instead of a whole run, while most tiles still save the extra row a run's last tile computes.
    const int bx = (int)blockIdx.x;
    const int lt0 = bx < Q.n_long ? bx * Q.run_long : Q.n_long * Q.run_long + (bx - Q.n_long);
    const int lt1_ = bx < Q.n_long ? lt0 + Q.run_long : lt0 + 1;
    const int lt1 = lt1_ < (int)P.ntiles ? lt1_ : (int)P.ntiles;       // (the host checks ntiles < 2^31: one scalar register each across the loops)
    if (lt0 == 0 && P.tile_lo == 0)                           // samples before the burst (row -1 of tile 0) are zeros: clear the ring
        for (int k = t; k < RSLOTS; k += MOD_THREADS) s_ring[k] = make_double2(0.0, 0.0);
    bool run_first = true;
    for (int ltile = lt0; ltile < lt1; ++ltile, run_first = false) {
        const bool run_last = ltile + 1 == lt1;
        // this tile's view of the arguments (see mcb_kargs): what only the tile prologue needs dies there
        auto kt = KA0;
        MCB_FRESH(kt);
        const auto &P = kt->P;
        const auto &Q = kt->Q;
        const int8_t *__restrict__ const symbols = kt->symbols;
        const double *__restrict__ const hvec = kt->hvec;
        double *__restrict__ const rows = kt->rows;
        // a window of a longer stream: local tile `ltile` is tile `tile` of the burst (P.tile_lo = 0,
        // P.ntiles = all of them for a one-shot burst); carries Wq are indexed locally, everything
        // else (symbols via mod_amp, samples, noise, columns) in burst coordinates
        const int64_t tile = P.tile_lo + ltile;
        const int64_t tile_base = tile * tile_len;
        const int64_t sym_base = tile_base / sps;
        const int cq_t = P.c / sps;                 // (= cq; per tile, so that it and what hangs on it are not carried across the row loop)
        const int64_t mp1_lo = sym_base + cq_t - JMAX + 1;
        const int lpart = JMAX - cq_t - P.dsh;      // window index of the first symbol not fully elapsed at the tile edge
        const bool full_tile = tile_base + tile_len <= P.out_len;
        // columns this tile may store, relative to its first symbol (32-bit tests per row)
        const int64_t klo64 = Q.k_lo - sym_base, khi64 = Q.k_hi - sym_base;
        const int klo = klo64 < -(1 << 20) ? -(1 << 20) : (int)klo64, khi = khi64 > (1 << 20) ? (1 << 20) : (int)khi64;
        // Packed rows (32 B per column): where column kr = 0 of this tile sits — a UNIFORM address (it may lie before the array:
        // columns outside [klo, khi) are never stored); a lane adds its own 32-bit byte offset, so a store is scalar base +
        // vector offset and the row loop does no 64-bit address arithmetic on the vector pipe (it did: 9 instructions per row).
        char *const tile_rows4 = reinterpret_cast<char *>(rows) + 32 * (sym_base - Q.k_lo);
        wf_lds_barrier();                                     // previous tile's columns are done with the ring and the window
        // The prologue's thread index is opaque per tile: otherwise every lane predicate and index of the
        // staging code below (t * 5 + e, lane == 63, t < 4, ...) is hoisted out of the tile loop as a
        // loop invariant and then SPILLED around the row loop (SGPR pairs for the masks, scratch for the indices).
        // (... and RE-DERIVED from the wave number and the lane's position in the wave: as a copy of t it kept t alive across the row loop)
        //  — from an opaque zero, or the derivation itself is hoisted out of the tile loop and carried)
        unsigned lane0 = 0u;
        asm volatile("" : "+v"(lane0));
        int tp = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, lane0)) + 64 * wave_u;
        mod_stage_window(symbols, hvec, P, mp1_lo - 1, win, s_amp, s_pi, s_wtot, tp);
        // noise index / 2 of this thread's samples in row 0 of the tile (tile_base and RS are even): the launch's first index
        // re-read per tile (carried, it was two more registers across the row loop) + the opaque thread index (as the loop
        // invariant "first index + t" it was carried, and spilled, across the row loop) — behind the window staging, whose
        // temporaries are the kernel's register peak
        const uint64_t pair_t = Q.pair0 + (Q.dyn_index ? (*Q.dyn_index >> 1) : 0ull) + (uint64_t)(tile_base >> 1) + (uint64_t)tp;
        const uint64_t *Wq = reinterpret_cast<const uint64_t *>(kt->scratch + MOD_OFF_P + P.ntiles);
        const double W = (double)Wq[ltile] * 0x1.0p-62 * P.sps_d;             // tile carry (tile 0: the head truncation -K0)
        const int ref_a = s_pi[lpart], ref_b = P.nh > 1 ? s_pi[win + 1 + lpart] : 0;
        // The first 8 samples of the NEXT tile — the last columns of row 15 look at them — are computed by the
        // run's last tile itself as "row 16" (row_step below: wave 0, the next tile's own carry; its window is
        // this one shifted by a tile's symbols, counts are differences: same integers, same expressions, so the
        // samples are bit-identical to the next tile's own row 0).  The last tile of a stream window is itself
        // halo (its last columns belong to the next chunk): zeros.
        // The ring is indexed by i' = (sample - d) mod 2048, d = offset of the window starts inside
        // the symbol grid: every window then starts on a pad-group boundary (8 samples + 1 pad slot),
        // so its 9 samples sit at CONSTANT slot offsets 0..7 and 9 from the group's first slot —
        // no per-tap index arithmetic.  Index 2048 (= 0 one turn later) has its own slot: the one
        // window that ends there reads the copy.
        // the 64 columns whose window starts in row `rho` of this tile (rho = -1 .. 15)
        auto bank_row = [&](int rho) __attribute__((always_inline)) {
            const int kr = CPR * rho + mq - Q.kshift;                   // column index relative to the tile's first symbol
            const int64_t k = sym_base + kr;
            const bool k_ok = kr >= klo && kr < khi && (CPR == 64 || mq < CPR);
            const int grp = ((rho & 1) ? CPR : 0) + (CPR == 64 ? mq : (mq < CPR ? mq : 0));   // pad group of the window start (rho = -1 .. 15)
            const bool odd = ((Q.pack_par0 + kr + (int)(sym_base & 1)) & 1) != 0;
            if constexpr (PAIRED && USE_MFMA) {
                // this wave's 16 symbols as in the 10-instruction form; A[i][4 kk + kq] = plane of tap 4 kk + kq of symbol i's window
                // (taps 0 .. 7 in slots 0 .. 7, tap 8 behind the pad slot, taps 9 .. 11 zeros): one 16-byte read feeds both tiles
                const int mqi = 32 * (wave_u >> 1) + 2 * mf_i + (wave_u & 1);
                const int grp_i = ((rho & 1) ? 64 : 0) + mqi;
                const double2 *xs = s_ring + 9 * grp_i + mf_kq;
                mcb_d4 accx = {0.0, 0.0, 0.0, 0.0}, accy = accx;
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    double2 a = kk < 2 ? xs[4 * kk] : xs[9 - mf_kq];            // (kk = 2: only tap 8 — kq = 0 — exists, in slot 9)
                    if (kk == 2 && mf_kq >= 1) a = make_double2(0.0, 0.0);
                    accx = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bmat[kk], accx, 0, 0, 0);
                    accy = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bmat[3 + kk], accy, 0, 0, 0);
                }
                // lane j: pair p = j & 7; j < 8: u = P, v = Q -> Re Z_p = u + v, Re Z_{15-p} = u - v; j >= 8: u = -S, v = R -> Im Z_p = u + v, Im Z_{15-p} = v - u
                const int pj = mf_i & 7, imj = mf_i >> 3;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int isym = mf_kq + 4 * reg;
                    const int kr_i = 64 * rho + 32 * (wave_u >> 1) + 2 * isym + (wave_u & 1) - Q.kshift;
                    const double u = accx[reg], v = accy[reg];
                    const double first = u + v, dd = u - v;
                    const double second = imj ? -dd : dd;
                    if (kr_i >= klo && kr_i < khi) {
                        double *o = rows + ((sym_base + kr_i) - Q.k_lo) * (2 * NF) + imj;
                        __builtin_nontemporal_store(first, &o[2 * pj]);
                        __builtin_nontemporal_store(second, &o[2 * (15 - pj)]);
                    }
                }
            } else if constexpr (USE_MFMA) {
                // this wave's 16 symbols: in-row column 32 (w >> 1) + 2 i + (w & 1); lane = (symbol i, k index kq)
                const int mqi = 32 * (wave_u >> 1) + 2 * mf_i + (wave_u & 1);
                const int grp_i = ((rho & 1) ? 64 : 0) + mqi;
                // A[i][4 kk + kq]: component c = 4 kk + kq of symbol i's window = (c & 1 ? Re : Im) of tap c >> 1
                const double *xa = ring_d + 2 * (9 * grp_i) + ((mf_kq & 1) ? 0 : 1) + 2 * (mf_kq >> 1);
                mcb_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) {
                    // taps 2 kk + (kq >> 1): kk < 4 -> slots 0 .. 7; kk = 4: tap 8 sits behind the pad slot (slot 9), taps 9 (c >= 18) are zeros
                    double a = kk < 4 ? xa[4 * kk] : xa[2 * 9 - 2 * (mf_kq >> 1)];
                    if (kk == 4 && mf_kq >= 2) a = 0.0;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bmat[kk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bmat[5 + kk], acc1, 0, 0, 0);
                }
                // D[i][j]: this lane holds output j = lane & 15 (+ 16 nb) of symbols i = (lane >> 4) + 4 reg
                const int jo = mf_i;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int isym = mf_kq + 4 * reg;
                    const int kr_i = 64 * rho + 32 * (wave_u >> 1) + 2 * isym + (wave_u & 1) - Q.kshift;
                    if (kr_i >= klo && kr_i < khi) {
                        double *o = rows + ((sym_base + kr_i) - Q.k_lo) * (2 * NF) + jo;
                        __builtin_nontemporal_store(acc0[reg], &o[0]);      // anyway — written normally they sat dirty in the L2s and their write-back met the
                        __builtin_nontemporal_store(acc1[reg], &o[16]);     // detector's first reads: same-box, detector 0.820 / 0.821 -> 0.754 / 0.761 ms
                    }
                }
            } else if constexpr (PAM2) {
                // A row i: operand row i & 7 (the samples from ring index RS (rho & 1) [- QSH SPS] + PG (i & 7)), plane i >> 3 (Re | Im);
                // this lane's element of k-step kk = 4 n + w is sample 4 kk + kq = 16 n + (4 w + kq): PG is a multiple of 16 and
                // 4 w + kq < 16, so its slot offset is the constant 16 n + 16 n / PG behind a per-lane base.
                const int ri = mf_i & 7;
                const int S0 = ((rho & 1) ? RS - GP::QSH * SPS : 0) + PG * (GP::OPROWS < 8 && ri >= GP::OPROWS ? 0 : ri);
                const double *xa = ring_d + 2 * (S0 + S0 / PG + 4 * wave_u + mf_kq) + (mf_i >> 3);
                mcb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int n = 0; n < GP::NKW; ++n)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2 * (16 * n + 16 * n / PG)], bpam[n], acc, 0, 0, 0);
                if (wave_u == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2 * (16 * GP::NKW + 16 * GP::NKW / PG)], bpam[GP::NKW], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) s_part[(4 * wave_u + reg) * 64 + (t & 63)] = acc[reg];
            } else if constexpr (PAM) {
                // operand row i = the 97 (121) samples from ring index RS (rho & 1) + PG i (odd rows of 51 columns: QSH
                // columns earlier); this lane's element of k-step kk = 4 n + w is component 4 kk + kq: sample
                // 8 n + 2 w + (kq >> 1), Im (kq even) or Re (kq odd).  The sample's slot is index + index / PG, the row
                // starts on a pad-group boundary, PG is a multiple of 8 and 2 w + (kq >> 1) < 8, so the slot offset of
                // step n is the CONSTANT 8 n + 8 n / PG behind a per-lane base.  (Operand rows past OPROWS: row 0 again.)
                const int S0 = ((rho & 1) ? RS - GP::QSH * SPS : 0) + PG * (GP::OPROWS < 16 && mf_i >= GP::OPROWS ? 0 : mf_i);
                const double *xa = ring_d + 2 * (S0 + S0 / PG + 2 * wave_u + (mf_kq >> 1)) + ((mf_kq & 1) ? 0 : 1);
                mcb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int n = 0; n < GP::NKW; ++n)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2 * (8 * n + 8 * n / PG)], bpam[n], acc, 0, 0, 0);
                if (wave_u == 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2 * (8 * GP::NKW + 8 * GP::NKW / PG)], bpam[GP::NKW], acc, 0, 0, 0);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) s_part[(4 * wave_u + reg) * 64 + (t & 63)] = acc[reg];
            } else if constexpr (PAIRED) {
                // Four filters as two conjugate pairs (PCM/FM: T[3 - q] == conj(T[q])): lane = (symbol, pair q = mp >> 1, plane
                // mp & 1).  The Re-plane lane runs P = sum x c and S = sum x s, the Im-plane lane Q = sum y s and R = sum y c — 18
                // multiply-adds instead of 36 —, one quad swap hands each lane the sum its partner holds for it, and
                //   Z_q = (P + Q) + j (R - S),   Z_{3-q} = (P - Q) + j (R + S):
                // the Re-plane lane stores the two real parts, the Im-plane lane the two imaginary parts.
                const int q = mp >> 1, pl = mp & 1;
                const double *xb = ring_d + 2 * (9 * grp) + pl;
                const int col = Q.cpm_nh == 2 ? (int)((sym_base + kr) & 1) : 0;
                const double2 *tb_ = s_taps + (col * NF + q) * 9;
                double sc = 0.0, ss = 0.0;                               // this plane against the cosine / sine taps
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const double xv = xb[2 * (j < 8 ? j : 9)];
                    const double2 tp = tb_[j];
                    sc = fma(xv, tp.x, sc);
                    ss = fma(xv, tp.y, ss);
                }
                // Re plane: keeps P = sc, sends S = ss;  Im plane: keeps R = sc, sends Q = ss
                const double got = mcb_quad_bcast<0xB1>(ss);             // quad_perm [1, 0, 3, 2]
                const double first = pl ? sc - got : sc + got;           // Re Z_q = P + Q  |  Im Z_q = R - S
                const double second = pl ? sc + got : sc - got;          // Re Z_{3-q} = P - Q  |  Im Z_{3-q} = R + S
                if (k_ok) {
                    double *o = rows + 2 * (k - Q.k_lo) * NF + pl;
                    __builtin_nontemporal_store(first, &o[2 * q]);
                    __builtin_nontemporal_store(second, &o[2 * (NF - 1 - q)]);
                }
            } else if constexpr (CPMNF != 0) {
                // rows of the generic CPM detector: this lane's CPMNF / 4 filters of symbol k
                constexpr int FPT = CPMNF / 4;
                const double2 *xb = s_ring + 9 * grp;
                const int col = Q.cpm_nh == 2 ? (int)((sym_base + kr) & 1) : 0;      // modulation-index column of symbol k
                const double2 *tb_ = s_taps + (col * CPMNF + FPT * mp) * 9;
                double zr[FPT], zi[FPT];
#pragma unroll
                for (int f = 0; f < FPT; ++f) zr[f] = zi[f] = 0.0;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const double2 xv = xb[j < 8 ? j : 9];
#pragma unroll
                    for (int f = 0; f < FPT; ++f) {
                        const double2 tp = tb_[f * 9 + j];
                        zr[f] = fma(xv.x, tp.x, fma(xv.y, tp.y, zr[f]));
                        zi[f] = fma(-xv.x, tp.y, fma(xv.y, tp.x, zi[f]));      // (imaginary sample's term first in both sums: cpm_oracle.c)
                    }
                }
                if (k_ok) {
                    double *o = rows + 2 * ((k - Q.k_lo) * CPMNF + FPT * mp);
#pragma unroll
                    for (int f = 0; f < FPT; ++f) {
                        __builtin_nontemporal_store(zr[f], &o[2 * f]);
                        __builtin_nontemporal_store(zi[f], &o[2 * f + 1]);
                    }
                }
            } else if (sym_taps_i != 0) {
                const double *xb = ring_d + 2 * (G::GS * grp) + (mp & 1);
                // (CPR even: the lane's column parity is the same in every row)
                const double *tapc = s_tapc + NT * (4 * ((CPR & 1) ? (int)odd : odd_l) + mp);
                const double x_first = xb[0];
                double S = x_first * tapc[0], U = x_first;                // (S = fma(x, tap, +0.0) and U = 0.0 + x are these values: one add less)
#pragma unroll NT <= 11 ? NT - 1 : 5                                     // (21 taps at sps 20: four trips of 5 keep the reads in flight within the register budget)
                for (int j = 1; j < NT; ++j) {
                    const double x = xb[2 * (j < SPS ? j : G::GS)];
                    S = fma(x, tapc[j], S);
                    U += x;
                }
                const double T = S + mcb_quad_bcast<0xB1>(S);            // quad_perm [1, 0, 3, 2]: vb | vb | va | va
                const double val = (mp == 0 || mp == 3) ? U : T;
                if (k_ok) {                                               // (32 B rows: the nontemporal hint measured neutral here)
                    *reinterpret_cast<double *>(tile_rows4 + (int64_t)(32 * (CPR * rho - Q.kshift)) + (unsigned)(32 * mq + 8 * slot_l)) = val;
                }
            } else {
                // any 3 x (SPS + 1) bank: lane p keeps the one chain of the packed row it stores
                const int f = mp < 2 ? 1 : (mp == 2 ? 0 : 2);
                const bool imag = mp == 1 || (mp == 2 && odd) || (mp == 3 && !odd);
                const double2 *xb = s_ring + G::GS * grp;
                double acc = 0.0;
#pragma unroll 1
                for (int j = 0; j < NT; ++j) {
                    const double2 x = xb[j < SPS ? j : G::GS];
                    const double2 tp = s_taps[f * NT + (NT - 1 - j)];
                    acc = imag ? fma(x.x, tp.y, fma(x.y, tp.x, acc)) : fma(x.x, tp.x, fma(-x.y, tp.y, acc));
                }
                if (k_ok) *reinterpret_cast<double *>(tile_rows4 + (int64_t)(32 * (CPR * rho - Q.kshift)) + (unsigned)(32 * mq + 8 * mp)) = acc;
            }
        };
        // PAM form: the partial tiles of bank_row(rho) summed and stored — register w of D is this wave's
        auto bank_reduce = [&](int rho) __attribute__((always_inline)) {
            if constexpr (PAM2) {
                // D[i][j] sits in lane j + 16 (i & 3), register i >> 2, i = operand row + 8 plane, j = 2 shift + filter.  This wave
                // stores tile columns 16 w .. 16 w + 15 (operand rows 2 w, 2 w + 1), a quad per column; lane p of the quad first
                // sums, over the four waves' partial tiles in the fixed order ((w0 + w1) + w2) + w3, component p of its column:
                // p = 0: Re b_0*r, 1: Im b_0*r, 2: Re b_1*r, 3: Im b_1*r — then forms packed slot p from the quad's four.
                const int ln = t & 63, tc = 16 * wave_u + (ln >> 2), p4 = ln & 3;
                const int ri = tc >> 3, reg = (ri >> 2) + 2 * (p4 & 1), lsrc = 2 * (tc & 7) + (p4 >> 1) + 16 * (ri & 3);
                const double *pp = s_part + reg * 64 + lsrc;
                const double v = ((pp[0] + pp[4 * 64]) + pp[8 * 64]) + pp[12 * 64];
                const double y0r = mcb_quad_bcast<0x00>(v), y0i = mcb_quad_bcast<0x55>(v), y1r = mcb_quad_bcast<0xAA>(v), y1i = mcb_quad_bcast<0xFF>(v);
                const int col = tc - ((rho & 1) ? GP::QSH : 0);
                const int kr = CPR * rho + col - Q.kshift;
                const int odd_c = (Q.pack_par0 + kr + (int)(sym_base & 1)) & 1;
                const double *gt = s_gt + 16 * odd_c + 4 * p4;
                const double out = fma(gt[3], y1i, fma(gt[2], y1r, fma(gt[1], y0i, gt[0] * y0r)));
                if (kr >= klo && kr < khi && (CPR == 64 || (col >= 0 && col < CPR)))
                    *reinterpret_cast<double *>(tile_rows4 + (int64_t)(32 * (CPR * rho - Q.kshift)) + (unsigned)(32 * col + 8 * p4)) = out;
            } else if constexpr (PAM) {
                const int ln = t & 63;
                const double *pp = s_part + (4 * 0 + wave_u) * 64 + ln;
                const double v = ((pp[0] + pp[4 * 64]) + pp[8 * 64]) + pp[12 * 64];
                // lane (j = 4 s + o, iq): column 16 w + 4 iq + s (- QSH in an odd row) of the row, packed slot o: 4 (4 iq + s) + o = the lane index
                const int col = 16 * wave_u + (ln >> 2) - ((rho & 1) ? GP::QSH : 0);
                const int kr = CPR * rho + col - Q.kshift;
                if (kr >= klo && kr < khi && (CPR == 64 || (col >= 0 && col < CPR))) rows[4 * ((sym_base + kr) - Q.k_lo) + (ln & 3)] = v;
            }
        };
        // One row: 512 samples by the 256 threads (EXTRA: "row 16", only what row 15's columns still need).
        auto row_step = [&](const int u, auto parc) __attribute__((always_inline)) {      // parc: u & 1 as a compile-time constant
            const bool EXTRA = u >= MOD_ROWS;                   // (uniform; false for every row but the run's last one)
            double2 x0 = make_double2(0.0, 0.0), x1 = x0;
            bool have_next = true;
            if ((!EXTRA || wave_u == 0) && active) {
            // row 16 runs on the NEXT tile's carry and reference counts (nothing of it is live across the other rows)
            auto next_tile_refs = [&](double &Wn, int &na, int &nb) __attribute__((always_inline)) {
                have_next = ltile + 1 < P.ntiles;
                Wn = have_next ? (double)Wq[ltile + 1] * 0x1.0p-62 * P.sps_d : 0.0;
                na = s_pi[lpart + MOD_ROWS * sym_per_row];
                nb = P.nh > 1 ? s_pi[win + 1 + lpart + MOD_ROWS * sym_per_row] : 0;
            };
            double ra, rb;
            double2 e0, e1;
            // The row's amplitude / count reads are ISSUED FIRST, the noise arithmetic (45 % of the row's vector
            // instructions, no dependence on them) runs behind them, the phase sums come last: left in program order
            // (reads, sums, sincos, then the noise) every wave started its row by waiting for LDS.  Same-box A/B at 1e7
            // symbols: 0.4717 / 0.4786 -> 0.4583 / 0.4531 ms (profiles/r03_ab_loads_first.log; with the three parts
            // pinned by sched_barrier or left to the scheduler in this source order: 0.4601 / 0.4574 against 0.4572 /
            // 0.4567 — it is the source order and the shared reads that count, so nothing is pinned; the NEXT row's
            // Philox block computed behind the second barrier, under the bank's reads: 0.4656 / 0.4700 against 0.4557 /
            // 0.4553, not kept).  (Even sps and even c —
            // every SOQPSK pulse of the reference at 8 / 10 / 20 samples per symbol: both samples of a pair see the same
            // symbols, 9 reads; wf_mod_chan_bank_applies admits nothing else there.  The 10-samples-per-symbol body keeps
            // the old order and the general pair form — MIL's 11 taps have c = 5 — and so do the CPM forms: ARTM's
            // measured 0.711 / 0.715 against 0.684 / 0.700 ms with the new order, PCM/FM's 24-tap pulse has an odd c.)
            // (10 samples per symbol: the long-pulse instantiation — SOQPSK-TG / -A / -B, 81 taps, c = 40 — takes this order too
            //  since the radius constants freed two registers: 128 exactly; the short-pulse one serves MIL's 11 taps, c = 5, and keeps the
            //  general pair form; the host admits an odd c at 10 samples per symbol only there.  The CPM forms measured 1.5 - 3 % slower
            //  with this order in round 3 and kept the old one; round 5, with the registers the shorter noise series freed: the paired
            //  16-filter form with a short pulse — ARTM — takes it, 372 -> 361 vector instructions per row, link 1.187 -> 1.179 ms same-box
            //  (profiles/r05_ab_artm_loads_first.log); the host sends an odd c to the unpaired form.  PCM/FM's c is odd.)
            constexpr bool LOADS_FIRST = (CPMNF <= 0 && (SPS != 10 || JMAX == 9)) || (CPMNF == 32 && JMAX <= 4);
            double am_[JMAX];
            int pi0_ = 0, pi1_ = 0;
            if constexpr (!LOADS_FIRST) {
                double Wu = W;
                int refu_a = ref_a, refu_b = ref_b;
                if (EXTRA) next_tile_refs(Wu, refu_a, refu_b);
                mod_pair_phase<JMAX>(Q0, Q1, &s_amp[l_top0p1 + u * sym_per_row], &s_pi[(q0 - cq) + u * sym_per_row], wrap, any_wrap, P.nh,
                                     win + 1, refu_a, refu_b, Wu, Th_a, Th_b, (double)SPS, 1.0 / (double)SPS, ra, rb);
                wf_sincos_sectors_pos(s_cis, fma(ra, sec_per_unit, sec_phi0), &e0.y, &e0.x);   // (ra, rb in [0, SPS), phi0 >= 0: the host checks)
                wf_sincos_sectors_pos(s_cis, fma(rb, sec_per_unit, sec_phi0), &e1.y, &e1.x);
            } else if constexpr (!(WF_MCB_SECTION_OFF & 4)) {
                const double *ap = &s_amp[l_top0p1 + u * sym_per_row];
                const int *pp = &s_pi[(q0 - cq) + u * sym_per_row];
#pragma unroll
                for (int j = 0; j < JMAX; ++j) am_[j] = ap[-1 - j];
                pi0_ = pp[0];
                if (P.nh > 1) pi1_ = pp[win + 1];
            }
            // channel (wf_awgn_c128): derotate + Philox AWGN, one block per thread and row
            double g[4];
            {
                // The Philox key schedule (20 words: seed + r * Weyl) is uniform and loop-invariant; left to
                // the compiler it is hoisted into 20 SGPRs and pushes as many other uniforms into spill
                // lanes (v_readlane reloads in this loop).  Opaque per row => re-derived by scalar adds.
                uint32_t k0 = (uint32_t)Q.seed, k1 = (uint32_t)(Q.seed >> 32);
                asm volatile("" : "+s"(k0), "+s"(k1));
                __builtin_amdgcn_s_setprio(0);
                // the two Box-Muller transforms interleaved, all four table entries fetched right behind the Philox rounds
                // (same-box A/B at 1e7 symbols: sequential 0.4580 / 0.4644 ms, table reads in pairs 0.4591 / 0.4588, all four up front 0.4479 / 0.4535;
                //  staged further — Philox + reads, then the phase sums and the sector reads, then the transforms — 0.4640 / 0.4598 against 0.4473 / 0.4536: not kept)
                // (the Philox counter of this thread's pair of samples: one 64-bit add per row on a per-tile, per-thread base)
                // (the one-operation form of (double)word + 1 where the registers allow: the sps-8 SOQPSK forms)
                wf_gaussian_two_il<true, decltype(tb), CPMNF == 0 && SPS == 8, WF_MCB_SECTION_OFF & 3>(pair_t + (uint64_t)(u * (RS / 2)), Q.stream_id, ((uint64_t)k1 << 32) | k0, Q.sigma, tb, g);
                __builtin_amdgcn_s_setprio(WF_MCB_PRIO);
            }
            if constexpr (LOADS_FIRST && (WF_MCB_SECTION_OFF & 4)) {
                e0 = make_double2(1.0, 0.0);                    // (section stub: see WF_MCB_SECTION_OFF)
                e1 = make_double2(0.0, 1.0);
            } else if constexpr (LOADS_FIRST) {
                // mod_pair_phase with both samples on the same symbols (a1 == a, one count)
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int j = 0; j < JMAX; ++j) {
                    acc0 = fma(Q0[j], am_[j], acc0);
                    acc1 = fma(Q1[j], am_[j], acc1);
                }
                // (the tile's carry and reference counts are used where they lie; row 16 fetches the next tile's HERE, into the
                //  same result: as copies selected at the top of every row they were three moves per row and four registers)
                double b0;
                if (!EXTRA) {
                    b0 = fma((double)(pi0_ - ref_a), Th_a, W);
                    if (P.nh > 1) b0 = fma((double)(pi1_ - ref_b), Th_b, b0);
                } else {
                    double Wn;
                    int na, nb;
                    next_tile_refs(Wn, na, nb);
                    b0 = fma((double)(pi0_ - na), Th_a, Wn);
                    if (P.nh > 1) b0 = fma((double)(pi1_ - nb), Th_b, b0);
                }
                const double v0 = b0 + acc0, v1 = b0 + acc1;
                const double kq = floor(v0 * (1.0 / (double)SPS));
                ra = fma(-kq, (double)SPS, v0);
                rb = fma(-kq, (double)SPS, v1);
                wf_sincos_sectors_pos(s_cis, fma(ra, sec_per_unit, sec_phi0), &e0.y, &e0.x);   // (ra, rb in [0, SPS), phi0 >= 0: the host checks)
                wf_sincos_sectors_pos(s_cis, fma(rb, sec_per_unit, sec_phi0), &e1.y, &e1.x);
            }
            x0 = make_double2(fma(e0.x, Q.rot_re, fma(-e0.y, Q.rot_im, g[0])), fma(e0.x, Q.rot_im, fma(e0.y, Q.rot_re, g[1])));
            x1 = make_double2(fma(e1.x, Q.rot_re, fma(-e1.y, Q.rot_im, g[2])), fma(e1.x, Q.rot_im, fma(e1.y, Q.rot_re, g[3])));
            if (EXTRA || !full_tile) {                          // (tile-uniform) samples past the end of the burst are zeros to the bank
                const int64_t n0 = tile_base + (int64_t)u * RS + 2 * t;
                if (n0 >= P.out_len || (EXTRA && !have_next)) x0 = make_double2(0.0, 0.0);
                if (n0 + 1 >= P.out_len || (EXTRA && !have_next)) x1 = make_double2(0.0, 0.0);
            }
            }
            // Two-row ring: row u takes the slots of row u - 2, whose last readers are the waves still in
            // bank_row(u - 2) ...
            wf_lds_barrier();
            // (PAM form) the partial tiles bank_row(u - 2) left during the previous row step
            if (PAM && u >= 1 && (u >= 2 || !run_first || (tile == 0 && Q.kshift < 0))) bank_reduce(u - 2);
            constexpr int XLEN = PAM ? GP::XLEN : SPS;          // samples of the next tile the last columns of a tile look at
            if constexpr (!RINGPK) {
                int ia, ib;                                     // ring indices (sample - d) mod RING of the thread's two samples
                if (SPS == 8) {
                    ia = ((u << 9) + 2 * t - Q.d) & (G::RING - 1);
                    ib = (ia + 1) & (G::RING - 1);
                } else {
                    ia = ((u & 1) ? RS : 0) + 2 * t - Q.d;
                    ia += ia < 0 ? G::RING : 0;
                    ib = ia + 1 == G::RING ? 0 : ia + 1;
                }
                if (active && (!EXTRA || 2 * t < XLEN)) {       // (row 16: its first XLEN samples; the lanes above computed on window slots that do not exist)
                    s_ring[ia + ia / PG] = x0;
                    s_ring[ib + ib / PG] = x1;
                    if constexpr (PAM) {
                    // indices 0 .. MIRROR-1 once more behind the ring (index RING + i)
                    if (ia < GP::MIRROR) s_ring[G::RING + ia + (G::RING + ia) / PG] = x0;
                    if (ib < GP::MIRROR) s_ring[G::RING + ib + (G::RING + ib) / PG] = x1;
                    } else {
                    if (ia == 0) s_ring[G::GROUPS * G::GS] = x0;
                    if (ib == 0) s_ring[G::GROUPS * G::GS] = x1;
                    }
                }
            } else if (active && (!EXTRA || 2 * t < XLEN)) {    // (row 16: its first XLEN samples; the lanes above computed on window slots that do not exist)
                constexpr int PAR = decltype(parc)::value;      // the row's parity (see ring_pk_a)
                *(mcb_lds2)(uintptr_t)(PAR ? ring_pk_a >> 16 : ring_pk_a & 0xFFFFu) = wf_v2d{x0.x, x0.y};
                *(mcb_lds2)(uintptr_t)(PAR ? ring_pk_b >> 16 : ring_pk_b & 0xFFFFu) = wf_v2d{x1.x, x1.y};
                if constexpr (PAR == 0) {                       // ring index 0 / RING - 1 | 0 live in an even row: the copy behind the ring
                    if (2 * t == Q.d) s_ring[G::GROUPS * G::GS] = x0;
                    if (2 * t + 1 == Q.d) s_ring[G::GROUPS * G::GS] = x1;
                }
            }
            // ... and row u is complete after this barrier — one more barrier per row than a four-row
            // ring, 18 KB less LDS (4 workgroups per CU instead of 3: 0.56 -> 0.5x ms).
            wf_lds_barrier();
            if constexpr (!(WF_MCB_SECTION_OFF & 8))
                if (u >= 1 || !run_first || (tile == 0 && Q.kshift < 0)) bank_row(u - 1);   // u = 0: the previous tile's row 15
        };
        // two rows per trip, ONE copy of the row code: row 16 goes through the same loop body (a separate
        // inlined copy — or the compiler's remainder loop — is a second 1200-instruction body whose
        // temporaries, added to everything live across the row loop, were what spilled)
        const int nrow = run_last ? MOD_ROWS + 1 : MOD_ROWS;
#pragma unroll 1
        for (int u = 0; u < nrow; u += 2) {
            row_step(u, std::integral_constant<int, 0>{});
            if (u + 1 < nrow) row_step(u + 1, std::integral_constant<int, 1>{});
        }
        if (PAM) {                                              // the last bank_row's partial tiles
            wf_lds_barrier();
            bank_reduce(nrow - 2);
        }
    }
#undef MCB_FRESH
}

// ------------------------------------------------------------------------------------------
// Modulator + channel, NOISY SAMPLES out (CPM link `fuse` bit 7, round 6): the front end of the link whose detector runs
// the matched filters itself (wf_cpm_lanes.hip, MF form).  A 16-filter row is 256 B per symbol, the 8 samples it is made
// from are 128 B: the ARTM link wrote 2.57 GB of rows per 1e7 symbols and read them 1.28 times.  This kernel is
// mod_main_kernel<JMAX, true> with the channel of mod_chan_bank_kernel applied to a thread's two samples before the
// store — the same expressions on the same operands in the same order as that kernel's row_step (mod_pair_phase,
// wf_sincos_sectors_pos, wf_gaussian_two_il, the two fma chains of the derotation), so the samples are bit for bit the
// ones its bank saw in LDS.  No ring, no bank, no workgroup barrier inside a tile.
struct mcs_kargs {
    const int8_t *symbols;
    const double *hvec, *pulse, *scratch;
    double *out;
    mod_params P;
    mcb_params Q;
};
typedef const __attribute__((address_space(4))) mcs_kargs *mcs_kptr;

template <int JMAX>
__global__ __launch_bounds__(MOD_THREADS, 4) void mod_chan_samples_kernel(const int8_t *__restrict__ symbols_, const double *__restrict__ hvec_,
                                                                          const double *__restrict__ pulse_, const double *__restrict__ scratch_,
                                                                          double *__restrict__ out_, mod_params P_, mcb_params Q_)
{
    mcs_kptr const KA0 = (mcs_kptr)__builtin_amdgcn_kernarg_segment_ptr();
    const auto &P = KA0->P;
    const double *__restrict__ const hvec = KA0->hvec;
    const double *__restrict__ const scratch = KA0->scratch;
    extern __shared__ double s_amp[];       // window of symbol amplitudes, then its prefix counts (ints)
    __shared__ int s_wtot[2 * MOD_WAVES];
    __shared__ double2 s_xp[2 * MOD_THREADS];   // wave-private transpose: pairs per lane -> rows of 64 samples
    __shared__ double2 s_tab[256];              // [0,128): log table, [128,256): sincos sectors
    const int t = threadIdx.x;
    wf_stage_tables<1, 0>(s_tab, t, MOD_THREADS);   // the tile loop starts with a barrier
    const double2 *s_cis = s_tab + 128;
    const wf_tabs_lds<1, 0> tb{s_tab};
    const int lane = t & 63, wave = t >> 6;
    const int sps = P.sps;
    const int tile_len = MOD_ROWS * P.rs;
    const int sym_per_row = P.rs / sps;
    const int cq = P.c / sps;
    const int q0 = (2 * t + P.c) / sps;
    const int r0 = (2 * t + P.c) - q0 * sps;
    const int wrap = (r0 + 1 == sps) ? 1 : 0;
    const int r1 = wrap ? 0 : r0 + 1;
    const bool any_wrap = (sps & 1) != 0 || (P.c & 1) != 0;
    const double *Gcum = scratch + MOD_OFF_GCUM(P.ntiles);
    double Q0[JMAX], Q1[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k0 = r0 + j * sps, k1 = r1 + j * sps;
        Q0[j] = Gcum[k0 < P.ntaps ? k0 : P.ntaps - 1];
        Q1[j] = Gcum[k1 < P.ntaps ? k1 : P.ntaps - 1];
    }
    const double sec_per_unit = 128.0 * P.inv_sps, sec_phi0 = 128.0 * P.phi0_turns;
    const int l_top0p1 = (q0 - cq) + JMAX;
    const int win = MOD_ROWS * sym_per_row + JMAX + 2;
    int *s_pi = reinterpret_cast<int *>(s_amp + ((win + 1) & ~1));
    const double T = scratch[0];
    const double Th_a = T * hvec[0], Th_b = P.nh > 1 ? T * hvec[1] : 0.0;
    const int lpart = JMAX - cq - P.dsh;

    for (int64_t tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        auto kt = KA0;
        asm volatile("" : "+s"(kt));
        const auto &P = kt->P;
        const auto &Q = kt->Q;
        const int8_t *__restrict__ const symbols = kt->symbols;
        const double *__restrict__ const hvec = kt->hvec;
        double *__restrict__ const out = kt->out;
        const uint64_t *Wq = reinterpret_cast<const uint64_t *>(kt->scratch + MOD_OFF_P + P.ntiles);
        int tp = t;
        asm volatile("" : "+v"(tp));
        const int64_t tile_g = P.tile_lo + tile;
        const int64_t tile_base = tile_g * tile_len;
        const int64_t sym_base = tile_base / sps;
        const int64_t mp1_lo = sym_base + cq - JMAX + 1;
        const bool full_tile = tile_base >= P.out_origin && tile_base + tile_len <= P.out_hi;
        wf_lds_barrier();
        mod_stage_window(symbols, hvec, P, mp1_lo - 1, win, s_amp, s_pi, s_wtot, tp);
        const uint64_t pair_t = Q.pair0 + (Q.dyn_index ? (*Q.dyn_index >> 1) : 0ull) + (uint64_t)(tile_base >> 1) + (uint64_t)tp;
        const double W = (double)Wq[tile] * 0x1.0p-62 * P.sps_d;
        const int ref_a = s_pi[lpart], ref_b = P.nh > 1 ? s_pi[win + 1 + lpart] : 0;
#pragma unroll 2
        for (int u = 0; u < MOD_ROWS; ++u) {
            double ra, rb;
            double2 e0, e1;
            mod_pair_phase<JMAX>(Q0, Q1, &s_amp[l_top0p1 + u * sym_per_row], &s_pi[(q0 - cq) + u * sym_per_row], wrap, any_wrap, P.nh > 1 ? 2 : 1,
                                 win + 1, ref_a, ref_b, W, Th_a, Th_b, P.sps_d, P.inv_sps, ra, rb);
            wf_sincos_sectors_pos(s_cis, fma(ra, sec_per_unit, sec_phi0), &e0.y, &e0.x);   // (ra, rb in [0, sps), phi0 >= 0: the host checks)
            wf_sincos_sectors_pos(s_cis, fma(rb, sec_per_unit, sec_phi0), &e1.y, &e1.x);
            double g[4];
            {
                uint32_t k0 = (uint32_t)Q.seed, k1 = (uint32_t)(Q.seed >> 32);
                asm volatile("" : "+s"(k0), "+s"(k1));
                wf_gaussian_two_il<true, decltype(tb), false, 0>(pair_t + (uint64_t)(u * (P.rs / 2)), Q.stream_id, ((uint64_t)k1 << 32) | k0, Q.sigma, tb, g);
            }
            const double2 x0 = make_double2(fma(e0.x, Q.rot_re, fma(-e0.y, Q.rot_im, g[0])), fma(e0.x, Q.rot_im, fma(e0.y, Q.rot_re, g[1])));
            const double2 x1 = make_double2(fma(e1.x, Q.rot_re, fma(-e1.y, Q.rot_im, g[2])), fma(e1.x, Q.rot_im, fma(e1.y, Q.rot_re, g[3])));
            // the wave's 128 samples through its private LDS strip: every store instruction writes 64 consecutive samples (mod_main_kernel)
            double2 *xw = s_xp + wave * (2 * WF_WAVE);
            xw[2 * lane] = x0;
            xw[2 * lane + 1] = x1;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double2 xa = xw[lane], xb = xw[WF_WAVE + lane];
            const int col = wave * (2 * WF_WAVE) + lane;
            const int64_t na = tile_base + (int64_t)u * P.rs + col, nb = na + WF_WAVE;
            double2 *o = reinterpret_cast<double2 *>(out);
            if (full_tile) {
                wf_store16_nt(o + (na - P.out_origin), xa);
                wf_store16_nt(o + (nb - P.out_origin), xb);
            } else {
                if (na >= P.out_origin && na < P.out_hi) wf_store16_nt(o + (na - P.out_origin), xa);
                if (nb >= P.out_origin && nb < P.out_hi) wf_store16_nt(o + (nb - P.out_origin), xb);
            }
            __builtin_amdgcn_wave_barrier();                    // (the strip is rewritten by the next row)
        }
    }
}

static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

// Geometry shared by the one-shot entry point and the streaming link.
static bool mod_setup(mod_params &P, int64_t nsym, int nh, int ntaps, int sps, double phi0, int max_nh = 2)
{
    // outside 2 <= sps <= 256 the row length below is 0 (or the lcm divides by 0): not a
    // configuration of the fused kernel.  P is left in a defined state for callers that read it.
    // (prefix counts: one set per modulation index — the one-kernel front ends keep two, the stand-alone modulator MOD_MAX_NH)
    if (sps < 2 || sps > 256 || ntaps < 1 || nsym < 1 || nh < 1 || nh > max_nh) {
        P = mod_params{};
        P.sps = sps;
        P.ntaps = ntaps;
        return false;
    }
    const int64_t npts = (nsym + 1) * (int64_t)sps;
    const int J = (ntaps + sps - 1) / sps;
    P.nsym = nsym;
    P.out_len = npts >= ntaps ? npts : ntaps;
    P.c = (int)(((npts >= ntaps ? (int64_t)ntaps : npts) - 1) / 2);
    P.sps = sps;
    P.ntaps = ntaps;
    P.nh = nh;
    const int l = sps / gcd_i(sps, 2) * 2;
    P.rs = (2 * MOD_THREADS / l) * l;
    const int64_t tile_len = (int64_t)MOD_ROWS * P.rs;
    P.spt = (int)(tile_len / sps);
    P.ntiles = (P.out_len + tile_len - 1) / tile_len;
    P.dsh = (ntaps - P.c + sps - 1) / sps;
    int R = (P.c - ntaps) % sps;
    if (R < 0) R += sps;
    P.npart = (ntaps - 1 + R) / sps;          // l = 0 .. npart-1 with idx_l >= 0
    P.nhead = P.c >= 1 ? (P.c - 1) / sps : 0;
    P.sps_d = (double)sps;
    P.inv_sps = 1.0 / (double)sps;
    P.phi0_turns = phi0 / (2.0 * M_PI);
    P.sym_origin = 0;
    P.nloc = nsym;
    P.tile_lo = 0;
    P.out_origin = 0;
    P.out_hi = P.out_len;
    P.q_out_tile = -1;
    // the analytic carry needs: pulse no longer than the signal, J within the register
    // budget, and every head / partial symbol of a tile edge inside the staged window
    return npts >= ntaps && J <= 33 && P.npart <= MOD_MAX_PART && P.spt >= P.dsh + J + P.nhead + 2;
}

// The two carry kernels (tile sums, tile scan) into the context's scratch.
// (slot 0 / 1: one of two sets of carries — a pipelined link computes the next block's while this block's main kernel reads its own)
static double *mod_scratch_of(wf_ctx *ctx, const mod_params &P, int slot)
{
    const size_t words = (MOD_OFF_GCUM(P.ntiles) + (size_t)P.ntaps + 31) / 32 * 32;
    return ctx->d_mod_scratch ? ctx->d_mod_scratch + (slot > 0 ? words : 0) : nullptr;
}
static int mod_launch_carries(wf_ctx *ctx, const mod_params &P, const int8_t *d_symbols, const double *d_h,
                              const double *d_pulse, const uint64_t *d_q_in, uint64_t *d_q_out, void *stream, int slot = -1)
{
    hipStream_t s = wf_stream(stream);
    const size_t words = (MOD_OFF_GCUM(P.ntiles) + (size_t)P.ntaps + 31) / 32 * 32;
    int rc = wf_ctx_reserve_mod(ctx, slot >= 0 ? 2 * words : words);
    if (rc) return rc;
    double *scratch = mod_scratch_of(ctx, P, slot);
    hipLaunchKernelGGL(mod_tile_sums_kernel, dim3((unsigned)((P.ntiles + MOD_WAVES - 1) / MOD_WAVES)), dim3(MOD_THREADS),
                       0, s, d_symbols, d_h, P, scratch);
    WF_LAUNCH_CHECK();
    hipLaunchKernelGGL(mod_tile_scan_kernel, dim3(1), dim3(1024), 0, s, d_symbols, d_h, d_pulse, P, scratch, d_q_in, d_q_out);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

static int mod_launch(wf_ctx *ctx, const mod_params &P, const int8_t *d_symbols, const double *d_h,
                      const double *d_pulse, double *d_out_ri, const uint64_t *d_q_in, uint64_t *d_q_out, void *stream)
{
    const int J = (P.ntaps + P.sps - 1) / P.sps;
    hipStream_t s = wf_stream(stream);
    int rc = mod_launch_carries(ctx, P, d_symbols, d_h, d_pulse, d_q_in, d_q_out, stream);
    if (rc) return rc;
    double *scratch = ctx->d_mod_scratch;
    const int64_t max_grid = 2048 * (256 / MOD_THREADS);
    const int grid = (int)(P.ntiles < max_grid ? P.ntiles : max_grid);
    const int sps = P.sps;
#define MOD_LAUNCH(JM)                                                                                   \
    do {                                                                                                 \
        const size_t win = (size_t)(MOD_ROWS * (P.rs / sps) + JM + 2);                                   \
        const size_t lds = ((win + 1) & ~(size_t)1) * sizeof(double) + (size_t)P.nh * (win + 1) * sizeof(int); \
        using kern_t = void (*)(const int8_t *, const double *, const double *, const double *, double *, mod_params); \
        const kern_t k = P.nh > 2 ? (P.rs == 2 * MOD_THREADS ? mod_main_kernel<JM, true, true> : mod_main_kernel<JM, false, true>) \
                                  : (P.rs == 2 * MOD_THREADS ? mod_main_kernel<JM, true> : mod_main_kernel<JM, false>); \
        if (lds > 48 * 1024)                                                                             \
            WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(MOD_THREADS), lds, s, d_symbols, d_h, d_pulse, scratch, d_out_ri, P); \
    } while (0)
    if (J <= 4) MOD_LAUNCH(4);
    else if (J <= 9) MOD_LAUNCH(9);
    else if (J <= 17) MOD_LAUNCH(17);
    else MOD_LAUNCH(33);
#undef MOD_LAUNCH
    WF_LAUNCH_CHECK();
    return WF_OK;
}

extern "C" int wf_cpm_modulate_c128(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h,
                                    int nh, const double *d_pulse, int ntaps, int sps, double phi0,
                                    double *d_out_ri, void *stream)
{
    WF_REQUIRE(ctx && d_h && d_pulse && d_out_ri, "wf_cpm_modulate_c128: NULL argument");
    WF_REQUIRE(nsym >= 0 && (nsym == 0 || d_symbols), "wf_cpm_modulate_c128: bad symbols");
    WF_REQUIRE(sps >= 2 || nsym == 0, "could not broadcast input array from shape (%lld,) into shape (%lld,)",
               (long long)nsym, (long long)(nsym - 1));
    WF_REQUIRE(sps >= 1 && sps <= 256 && nh >= 1 && ntaps >= 1, "wf_cpm_modulate_c128: sps %d nh %d ntaps %d", sps, nh, ntaps);
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0, "wf_cpm_modulate_c128: d_out alignment");
    WF_HIP(hipSetDevice(ctx->device));
    mod_params P;
    if (!mod_setup(P, nsym, nh, ntaps, sps, phi0, MOD_MAX_NH)) return 1;  // caller falls back to the two stage kernels
    return mod_launch(ctx, P, d_symbols, d_h, d_pulse, d_out_ri, nullptr, nullptr, stream);
}

// Fused modulator + channel + 3 x 9 bank with detector-packed rows (link fuse bit 3; internal).
// Returns 1 — not an error — when the configuration is outside the fused kernel (the caller then
// runs the separate kernels): needs sps 8, 10 or 20 (CPM rows: 8), a pulse of at most 9 symbols inside the fused
// modulator's envelope, an even first noise index.
// Window form (streaming link): tiles [tile_lo, tile_lo + ntiles) of a burst of nsym_total symbols,
// d_symbols[0] = symbol sym_origin (nloc resident), phase carry of the first tile from *d_q_in
// (ignored for tile_lo == 0), carry of local tile q_out_tile to *d_q_out, columns [k_lo, k_lo + ncols)
// stored at d_rows4[4 (k - k_lo)].  One-shot: tile_lo = 0, ntiles = -1 (all), k_lo = 0.
int wf_mod_chan_bank_window(wf_ctx *ctx, const int8_t *d_symbols, int64_t sym_origin, int64_t nloc, int64_t nsym_total,
                            const double *d_h, int nh, const double *d_pulse, int ntaps, int sps, double phi0,
                            int64_t tile_lo, int64_t ntiles, const uint64_t *d_q_in, uint64_t *d_q_out, int64_t q_out_tile,
                            const double *d_mf_taps, double rot_re, double rot_im, double sigma, uint64_t seed,
                            uint64_t stream_id, uint64_t first_index, const uint64_t *d_dyn_index, int64_t first,
                            int64_t k_lo, int64_t ncols, int pack_par0, double *d_rows4, void *stream, int cpm_nf, int cpm_nh,
                            int stage, int mf_ntaps, const wf_mcb_opts *opts)
{
    const wf_mcb_opts none{};
    const wf_mcb_opts &O = opts ? *opts : none;
    // stage 1: the two carry kernels only (they also export the phase carry of a stream window);
    // stage 2: the main kernel only, on carries an earlier stage-1 call left in this context's
    // scratch; 3: both.  Callers that pipeline chunks split them so that the next chunk's carries do
    // not wait for this chunk's main kernel.
    WF_REQUIRE(ctx && d_symbols && d_h && d_pulse && d_mf_taps && d_rows4, "wf_mod_chan_bank: NULL argument");
    if (cpm_nf != 0 && cpm_nf != 4 && cpm_nf != 16) return 1;
    if (cpm_nf && cpm_nh != 1 && cpm_nh != 2) return 1;
    // (the kernel's sector split — wf_sincos_sectors_pos: index (int)y, remainder fract(y) — takes the phase offset to be at least ONE
    //  sector, 2 pi / 128: the phase reduced mod sps can come out a rounding error below 0 at 10 / 20 samples per symbol, where 1 / sps
    //  is inexact, and with no offset the argument would then be negative — truncation is not floor there.  Every caller passes pi / 4.)
    if (!(phi0 * (128.0 / (2.0 * M_PI)) >= 1.0)) return 1;
    // SOQPSK bank: sps + 1 taps = the pulse-truncation form; any other odd length up to MCB_PAM_NT at 8 samples per
    // symbol = the long-bank (PAM) form on the matrix cores
    if (mf_ntaps <= 0) mf_ntaps = sps + 1;
    const bool pam = cpm_nf == 0 && mf_ntaps != sps + 1;
    if (pam && ((sps != 8 && sps != 10) || mf_ntaps < 3 || mf_ntaps > MCB_PAM_NT(sps) || (mf_ntaps & 1) == 0)) return 1;
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_rows4) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_mf_taps) & 15) == 0,
               "wf_mod_chan_bank: device pointers must be 16-byte aligned");
    mod_params P;
    const bool sps_ok = sps == 8 || (cpm_nf == 0 && (sps == 10 || sps == 20));   // kernel instantiations (mcb_geom)
    if (!sps_ok || (first_index & 1) || first < 0 || first >= sps || ncols < 1 || k_lo < 0) return 1;
    const int rs_want = sps == 8 ? mcb_geom<8>::RS : (sps == 10 ? mcb_geom<10>::RS : mcb_geom<20>::RS);
    if (!mod_setup(P, nsym_total, nh, ntaps, sps, phi0) || P.rs != rs_want) return 1;
    const int J = (ntaps + sps - 1) / sps;
    if (J > 9) return 1;
    if (cpm_nf == 0 && (sps != 10 || J > 4) && ((sps & 1) || (P.c & 1))) return 1;     // SOQPSK forms at 8 / 20, and the long-pulse form at 10: both samples of a thread's pair under the same symbols
    WF_REQUIRE(first + (k_lo + ncols - 1) * (int64_t)sps < P.out_len, "wf_mod_chan_bank: columns run past the burst");
    if (ntiles < 0) ntiles = P.ntiles;
    WF_REQUIRE(tile_lo >= 0 && ntiles >= 1 && tile_lo + ntiles <= P.ntiles && ntiles < (int64_t)1 << 31, "wf_mod_chan_bank: bad tile window");
    P.sym_origin = sym_origin;
    P.nloc = nloc;
    P.tile_lo = tile_lo;
    P.ntiles = ntiles;
    P.q_out_tile = q_out_tile;
    WF_HIP(hipSetDevice(ctx->device));
    if (stage & 1) {
        int rc = mod_launch_carries(ctx, P, d_symbols, d_h, d_pulse, d_q_in, d_q_out, stream, O.scratch_slot);
        if (rc) return rc;
    }
    if (!(stage & 2)) return WF_OK;
    WF_REQUIRE(ctx->d_mod_scratch != nullptr, "wf_mod_chan_bank: stage 2 without the carries of stage 1");
    double *const scratch_main = mod_scratch_of(ctx, P, O.scratch_slot);
    mcb_params Q;
    Q.rot_re = rot_re; Q.rot_im = rot_im; Q.sigma = sigma;
    Q.seed = seed; Q.stream_id = stream_id; Q.pair0 = first_index >> 1;
    Q.dyn_index = d_dyn_index;
    Q.k_lo = k_lo;
    Q.k_hi = k_lo + ncols;
    // column k's window (sps + 1 taps, "same" convolution: centre tap at sample first + k sps) starts at sample
    // first + k sps - sps / 2 = sps (k + kshift) + d
    // (any odd bank length: centre tap (mf_ntaps - 1) / 2, window start = first + k sps - (mf_ntaps - 1) / 2)
    {
        const int64_t off = first - (cpm_nf ? sps / 2 : (mf_ntaps - 1) / 2);
        int64_t ks = off / sps;
        if (off - ks * sps < 0) --ks;
        Q.kshift = (int)ks;
        Q.d = (int)(off - ks * sps);
    }
    Q.mf_ntaps = mf_ntaps;
    Q.pack_par0 = pack_par0 & 1;
    Q.cpm_nh = cpm_nh;
    const int JM = J <= 4 ? 4 : 9;
    const int win = MOD_ROWS * (P.rs / sps) + JM + 2 + (pam ? 10 : 0);
    const int ring_slots = sps == 8 ? mcb_geom<8>::SLOTS : (sps == 10 ? mcb_geom<10>::SLOTS : mcb_geom<20>::SLOTS);
    // (occupancy experiment, LDS padded to force fewer workgroups per CU with the 4-row ring of the
    //  first version: 1 per CU 0.97 ms, 2: 0.63, 3: 0.56 — the 2-row ring's 4 per CU: 0.53)
    // the long bank in factored form (the caller's link handed the factorisation over): the kernel then reads ITS buffer as mf_taps
    const bool pam2 = pam && O.pam_factor != nullptr;
    if (pam2) {
        WF_REQUIRE((reinterpret_cast<uintptr_t>(O.pam_factor) & 7) == 0, "wf_mod_chan_bank: the bank's factorisation must be 8-byte aligned");
        d_mf_taps = O.pam_factor;
    }
    const size_t lds = (size_t)((win + 1) & ~1) * sizeof(double) + (size_t)(pam ? mcb_pam_slots(sps, pam2) : ring_slots) * sizeof(double2) +
                       (size_t)nh * (win + 1) * sizeof(int);
    // one run of consecutive tiles per resident workgroup (4 per CU for the SOQPSK form, 3 for the CPM forms)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
#ifndef WF_MCB_TAIL_SLOTS_DEFAULT
#define WF_MCB_TAIL_SLOTS_DEFAULT 1.5   // tiles that go out ONE per workgroup at the end of a launch, in resident-slot-fulls (cus x workgroups per CU):
#endif                                   // same-box, 1e7 symbols, two passes: 0: 0.4455 / 0.4522 ms | 0.5: 0.4500 / 0.4525 | 1: 0.4436 / 0.4384 | 1.5: 0.4366 / 0.4383 | 2: 0.4374 / 0.4517 | 3: 0.4426 / 0.4530
#ifndef WF_MCB_RUNS_PER_SLOT
#define WF_MCB_RUNS_PER_SLOT 4   // same-box A/B at 1e7 symbols: 1 run per slot 0.517 ms, 2: 0.500, 4: 0.478, 5 / 10: 0.486 (finer runs balance better; longer ones save more halo rows)
#endif
    // (with the single-tile tail below the SOQPSK forms do better on longer runs: 2 per slot = 5 tiles at 1e7 symbols, 0.4352 / 0.4345 ms
    //  against 0.4418 / 0.4427 for 4 per slot, 0.487 for 1, 0.4515 / 0.4424 for 3)
#ifndef WF_MCB_RUNS_PER_SLOT_SOQPSK
#define WF_MCB_RUNS_PER_SLOT_SOQPSK 2
#endif
    // (a pipelined CPM link asks for finer runs, wf_mcb_opts::runs_hint: its detector runs BESIDE this kernel and gets its
    //  waves onto a SIMD only when one of this kernel's workgroups leaves.  Steady state of the pipelined links, same box,
    //  runs per slot 1 | 2 | 4 | 8 | 16: PCM/FM 0.940 | 0.856 | 0.817 | 0.806 | 0.800 ms, ARTM 1.481 | 1.348 | 1.330 | 1.324 | 1.330)
    const int runs_per_slot = cpm_nf ? (O.runs_hint > 0 ? O.runs_hint : WF_MCB_RUNS_PER_SLOT) : WF_MCB_RUNS_PER_SLOT_SOQPSK;
    // (round 4, negative: a pipelined CPM link's front end capped at 2 | 2.5 | 3 | 4 | 8 | 16 resident-slot-fulls of workgroups, to
    //  leave each CU half free for the detector running beside it: ARTM steady 1.87 | 1.69 | 1.53 | 1.50 | 1.35 | 1.31 ms against
    //  1.32 uncapped, PCM/FM 0.74 | 0.91 | 0.85 | 0.74 | 0.70 | 0.67 against 0.65 — this kernel needs its four waves per SIMD;
    //  profiles/r04_ab_front_end_grid_cap.log)
    const int64_t max_grid = (int64_t)cus * ((cpm_nf && JM != 4) || pam ? 3 : 4) * runs_per_slot;
    const int64_t per_run = (P.ntiles + max_grid - 1) / max_grid;
    // the last `tail` tiles go out one per workgroup (WF_OPT_MCB_TAIL_PERMILLE / 1000 resident-slot-fulls of them)
    const int64_t tail_opt = ctx->opt[WF_OPT_MCB_TAIL_PERMILLE];        // (a tuning aid: 0 = the default above, -1 = equal runs throughout)
    const double tail_slots = tail_opt == 0 ? WF_MCB_TAIL_SLOTS_DEFAULT : (tail_opt < 0 ? 0.0 : (double)tail_opt / 1000.0);
    // (SOQPSK forms only: PT -1.9 %, PAM -3.8 %; the ARTM form, whose 2.56 GB of row stores bind it, measured + 1.3 % and its detector + 1.5 %)
    int64_t tail = per_run > 1 && cpm_nf == 0 ? (int64_t)(tail_slots * (double)(max_grid / runs_per_slot)) : 0;
    if (tail > P.ntiles / 2) tail = P.ntiles / 2;
    Q.run_long = (int)per_run;
    Q.n_long = (int)((P.ntiles - tail) / per_run);
    const int grid = (int)(Q.n_long + (P.ntiles - (int64_t)Q.n_long * per_run));
    using kern_t = void (*)(const int8_t *, const double *, const double *, const double *, const double *, double *, mod_params, mcb_params);
    // (16 templates that pair off as conjugates — the link checked it —, f <-> 15 - f: the four-real-sums form, 6 matrix instructions per 16 symbols for 10)
    // (<4, 32> reads a pair's symbols once for both samples: even c only — an odd c takes the unpaired form, which is right for any templates)
    kern_t k = cpm_nf == 16 && O.cpm_paired && (JM != 4 || (P.c & 1) == 0) ? (JM == 4 ? mod_chan_bank_kernel<4, 32> : mod_chan_bank_kernel<9, 32>)
             : cpm_nf == 16 ? (JM == 4 ? mod_chan_bank_kernel<4, 16> : mod_chan_bank_kernel<9, 16>)
             : cpm_nf == 4 && O.cpm_paired ? (JM == 4 ? mod_chan_bank_kernel<4, 8> : mod_chan_bank_kernel<9, 8>)
             : cpm_nf == 4  ? (JM == 4 ? mod_chan_bank_kernel<4, 4> : mod_chan_bank_kernel<9, 4>)
             : pam2         ? (sps == 10 ? (JM == 4 ? mod_chan_bank_kernel<4, -2, 10> : mod_chan_bank_kernel<9, -2, 10>)
                                         : (JM == 4 ? mod_chan_bank_kernel<4, -2> : mod_chan_bank_kernel<9, -2>))
             : pam          ? (sps == 10 ? (JM == 4 ? mod_chan_bank_kernel<4, -1, 10> : mod_chan_bank_kernel<9, -1, 10>)
                                         : (JM == 4 ? mod_chan_bank_kernel<4, -1> : mod_chan_bank_kernel<9, -1>))
             : sps == 10    ? (JM == 4 ? mod_chan_bank_kernel<4, 0, 10> : mod_chan_bank_kernel<9, 0, 10>)
             : sps == 20    ? (JM == 4 ? mod_chan_bank_kernel<4, 0, 20> : mod_chan_bank_kernel<9, 0, 20>)
                            : (JM == 4 ? mod_chan_bank_kernel<4, 0> : mod_chan_bank_kernel<9, 0>);
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(MOD_THREADS), lds, wf_stream(stream), d_symbols, d_h, d_pulse, scratch_main, d_mf_taps,
                       d_rows4, P, Q);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

int wf_mod_chan_bank_packed(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h, int nh,
                            const double *d_pulse, int ntaps, int sps, double phi0, const double *d_mf_taps, double rot_re,
                            double rot_im, double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                            int64_t first, int64_t ncols, int pack_par0, double *d_rows4, void *stream, int mf_ntaps, const wf_mcb_opts *opts,
                            int stage)
{
    return wf_mod_chan_bank_window(ctx, d_symbols, 0, nsym, nsym, d_h, nh, d_pulse, ntaps, sps, phi0, 0, -1, nullptr, nullptr, -1,
                                   d_mf_taps, rot_re, rot_im, sigma, seed, stream_id, first_index, nullptr, first, 0, ncols,
                                   pack_par0, d_rows4, stream, 0, 1, stage, mf_ntaps, opts);
}

// Would the one-kernel front end (SOQPSK form: 3 x (sps + 1) bank, detector-packed rows) take this configuration?
// Exactly the conditions under which wf_mod_chan_bank_window returns 0, without launching anything: the link
// decides the row layout (32 B packed at sps 10 / 20 only through this kernel) with it.
int wf_mod_chan_bank_applies(int64_t nsym, int nh, int ntaps, int sps, int mf_ntaps, int64_t first)
{
    const bool pam = (sps == 8 || sps == 10) && mf_ntaps != sps + 1 && mf_ntaps >= 3 && mf_ntaps <= MCB_PAM_NT(sps) && (mf_ntaps & 1);   // long-bank form
    if (!(sps == 8 || sps == 10 || sps == 20) || (mf_ntaps != sps + 1 && !pam) || first < 0 || first >= sps || nh < 1 || nh > 2) return 0;
    mod_params P;
    const int rs_want = sps == 8 ? mcb_geom<8>::RS : (sps == 10 ? mcb_geom<10>::RS : mcb_geom<20>::RS);
    if (!mod_setup(P, nsym, nh, ntaps, sps, 0.0) || P.rs != rs_want) return 0;
    if ((sps != 10 || (ntaps + sps - 1) / sps > 4) && ((sps & 1) || (P.c & 1))) return 0;      // (only the short-pulse body at 10 samples per symbol keeps the general pair form)
    return (ntaps + sps - 1) / sps <= 9;
}

// Would the one-kernel front end take this CPM configuration?  (The answer wf_mod_chan_cpm_rows gives by
// returning 0 / 1, without launching anything: wf_cpm_link_layout reports it, so that callers need not infer
// the path from stage times.)
int wf_mod_chan_cpm_rows_applies(int64_t nsym, int nh, int ntaps, int sps, int nfilt, int ntm, int64_t start0)
{
    if (ntm != 9 || start0 < -4 || start0 > 3 || (nfilt != 4 && nfilt != 16) || sps != 8 || (nh != 1 && nh != 2)) return 0;
    mod_params P;
    if (!mod_setup(P, nsym, nh, ntaps, sps, 0.0) || P.rs != 2 * MOD_THREADS) return 0;
    return (ntaps + sps - 1) / sps <= 9;
}

// The same kernel producing the generic CPM detector's matched-filter rows (nfilt = 4 or 16 templates
// of 9 taps per modulation-index column; row n from samples [start0 + 8 n, + 8]).  Returns 1 when
// outside the kernel (caller runs modulator, channel and wf_cpm_mf_rows_c128 separately).
int wf_mod_chan_cpm_rows(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h, int nh, const double *d_pulse,
                         int ntaps, int sps, double phi0, const double *d_templates, int nfilt, int ntm, int64_t start0,
                         double rot_re, double rot_im, double sigma, uint64_t seed, uint64_t stream_id, int64_t ncalls,
                         double *d_rows, void *stream, const wf_mcb_opts *opts)
{
    if (ntm != 9 || start0 < -4 || start0 > 3 || (nfilt != 4 && nfilt != 16)) return 1;
    return wf_mod_chan_bank_window(ctx, d_symbols, 0, nsym, nsym, d_h, nh, d_pulse, ntaps, sps, phi0, 0, -1, nullptr, nullptr, -1,
                                   d_templates, rot_re, rot_im, sigma, seed, stream_id, 0, nullptr, start0 + 4, 0, ncalls, 0,
                                   d_rows, stream, nfilt, nh, 3, 0, opts);
}

// Modulator + channel with the noisy samples stored (mod_chan_samples_kernel): d_out_ri[n] = sample n of the burst, n < (nsym + 1) sps.
// Would it take this configuration?  (rows of exactly 512 samples, at most two modulation indices, a pulse of at most 9 symbols)
int wf_mod_chan_samples_applies(int64_t nsym, int nh, int ntaps, int sps)
{
    mod_params P;
    if (nh < 1 || nh > 2 || !mod_setup(P, nsym, nh, ntaps, sps, 0.0) || P.rs != 2 * MOD_THREADS) return 0;
    return (ntaps + sps - 1) / sps <= 9;
}

// Returns 1 — not an error — when the configuration is outside the kernel (the caller runs modulator and channel separately).
int wf_mod_chan_samples(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h, int nh, const double *d_pulse, int ntaps,
                        int sps, double phi0, double rot_re, double rot_im, double sigma, uint64_t seed, uint64_t stream_id,
                        uint64_t first_index, double *d_out_ri, void *stream)
{
    WF_REQUIRE(ctx && d_symbols && d_h && d_pulse && d_out_ri, "wf_mod_chan_samples: NULL argument");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0, "wf_mod_chan_samples: d_out alignment");
    if (!(phi0 * (128.0 / (2.0 * M_PI)) >= 1.0) || (first_index & 1) || !wf_mod_chan_samples_applies(nsym, nh, ntaps, sps)) return 1;   // (one sector of margin: wf_mod_chan_bank_window)
    mod_params P;
    if (!mod_setup(P, nsym, nh, ntaps, sps, phi0)) return 1;
    WF_HIP(hipSetDevice(ctx->device));
    int rc = mod_launch_carries(ctx, P, d_symbols, d_h, d_pulse, nullptr, nullptr, stream);
    if (rc) return rc;
    mcb_params Q{};
    Q.rot_re = rot_re; Q.rot_im = rot_im; Q.sigma = sigma;
    Q.seed = seed; Q.stream_id = stream_id; Q.pair0 = first_index >> 1;
    Q.dyn_index = nullptr;
    const int J = (ntaps + sps - 1) / sps;
    const int JM = J <= 4 ? 4 : 9;
    const size_t win = (size_t)(MOD_ROWS * (P.rs / sps) + JM + 2);
    const size_t lds = ((win + 1) & ~(size_t)1) * sizeof(double) + (size_t)P.nh * (win + 1) * sizeof(int);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int64_t max_grid = (int64_t)cus * 4 * 4;              // (grid-stride over the tiles: a few rounds of resident workgroups)
    const int grid = (int)(P.ntiles < max_grid ? P.ntiles : max_grid);
    using kern_t = void (*)(const int8_t *, const double *, const double *, const double *, double *, mod_params, mcb_params);
    const kern_t k = JM == 4 ? mod_chan_samples_kernel<4> : mod_chan_samples_kernel<9>;
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(MOD_THREADS), lds, wf_stream(stream), d_symbols, d_h, d_pulse, ctx->d_mod_scratch, d_out_ri, P, Q);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// Window form of wf_mod_chan_samples (the streaming CPM link): tiles [tile_lo, tile_lo + ntiles) of a burst of nsym_total symbols,
// d_symbols[0] = symbol sym_origin (nloc resident), phase carry of the first tile from *d_q_in (ignored for tile_lo == 0), carry of
// local tile q_out_tile to *d_q_out; d_out_ri[0] is global sample out_origin and samples [out_origin, out_hi) are stored (the tiles may
// cover more).  The noise of sample n is that of absolute index first_index + n whatever the window.
int wf_mod_chan_samples_window(wf_ctx *ctx, const int8_t *d_symbols, int64_t sym_origin, int64_t nloc, int64_t nsym_total, const double *d_h, int nh,
                               const double *d_pulse, int ntaps, int sps, double phi0, int64_t tile_lo, int64_t ntiles, const uint64_t *d_q_in,
                               uint64_t *d_q_out, int64_t q_out_tile, double rot_re, double rot_im, double sigma, uint64_t seed, uint64_t stream_id,
                               uint64_t first_index, double *d_out_ri, int64_t out_origin, int64_t out_hi, void *stream)
{
    WF_REQUIRE(ctx && d_symbols && d_h && d_pulse && d_out_ri, "wf_mod_chan_samples_window: NULL argument");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0, "wf_mod_chan_samples_window: d_out alignment");
    if (!(phi0 * (128.0 / (2.0 * M_PI)) >= 1.0) || (first_index & 1) || !wf_mod_chan_samples_applies(nsym_total, nh, ntaps, sps)) return 1;
    mod_params P;
    if (!mod_setup(P, nsym_total, nh, ntaps, sps, phi0)) return 1;
    const int64_t tile_len = (int64_t)MOD_ROWS * P.rs;
    WF_REQUIRE(tile_lo >= 0 && ntiles >= 1 && tile_lo + ntiles <= P.ntiles && out_origin >= tile_lo * tile_len && out_hi >= out_origin,
               "wf_mod_chan_samples_window: bad tile window");
    P.sym_origin = sym_origin;
    P.nloc = nloc;
    P.tile_lo = tile_lo;
    P.ntiles = ntiles;
    P.out_origin = out_origin;
    const int64_t hi = (tile_lo + ntiles) * tile_len;
    P.out_hi = out_hi < hi ? out_hi : hi;
    if (P.out_hi > P.out_len) P.out_hi = P.out_len;
    P.q_out_tile = q_out_tile;
    WF_HIP(hipSetDevice(ctx->device));
    int rc = mod_launch_carries(ctx, P, d_symbols, d_h, d_pulse, d_q_in, d_q_out, stream);
    if (rc) return rc;
    mcb_params Q{};
    Q.rot_re = rot_re; Q.rot_im = rot_im; Q.sigma = sigma;
    Q.seed = seed; Q.stream_id = stream_id; Q.pair0 = first_index >> 1;
    Q.dyn_index = nullptr;
    const int J = (ntaps + sps - 1) / sps;
    const int JM = J <= 4 ? 4 : 9;
    const size_t win = (size_t)(MOD_ROWS * (P.rs / sps) + JM + 2);
    const size_t lds = ((win + 1) & ~(size_t)1) * sizeof(double) + (size_t)P.nh * (win + 1) * sizeof(int);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const int64_t max_grid = (int64_t)cus * 4 * 4;
    const int grid = (int)(P.ntiles < max_grid ? P.ntiles : max_grid);
    using kern_t = void (*)(const int8_t *, const double *, const double *, const double *, double *, mod_params, mcb_params);
    const kern_t k = JM == 4 ? mod_chan_samples_kernel<4> : mod_chan_samples_kernel<9>;
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(MOD_THREADS), lds, wf_stream(stream), d_symbols, d_h, d_pulse, ctx->d_mod_scratch, d_out_ri, P, Q);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// Streaming form (internal): modulate the tiles [tile_lo, tile_lo + ntiles) of a burst of nsym_total
// symbols.  d_symbols[0] is global symbol sym_origin (nloc resident), d_out_ri[0] is global sample
// out_origin.  The phase carry of the window's first tile comes from *d_q_in (ignored when tile_lo == 0)
// and the carry of local tile q_out_tile goes to *d_q_out (same word allowed).
int wf_cpm_modulate_window(wf_ctx *ctx, const int8_t *d_symbols, int64_t sym_origin, int64_t nloc, int64_t nsym_total,
                           const double *d_h, int nh, const double *d_pulse, int ntaps, int sps, double phi0,
                           int64_t tile_lo, int64_t ntiles, double *d_out_ri, int64_t out_origin,
                           const uint64_t *d_q_in, uint64_t *d_q_out, int64_t q_out_tile, void *stream)
{
    WF_HIP(hipSetDevice(ctx->device));
    mod_params P;
    WF_REQUIRE(mod_setup(P, nsym_total, nh, ntaps, sps, phi0), "wf_cpm_modulate_window: configuration outside the fused kernel");
    const int64_t tile_len = (int64_t)MOD_ROWS * P.rs;
    WF_REQUIRE(tile_lo >= 0 && ntiles >= 1 && tile_lo + ntiles <= P.ntiles && out_origin == tile_lo * tile_len,
               "wf_cpm_modulate_window: bad tile window");
    P.sym_origin = sym_origin;
    P.nloc = nloc;
    P.tile_lo = tile_lo;
    P.ntiles = ntiles;
    P.out_origin = out_origin;
    const int64_t hi = (tile_lo + ntiles) * tile_len;
    P.out_hi = hi < P.out_len ? hi : P.out_len;
    P.q_out_tile = q_out_tile;
    return mod_launch(ctx, P, d_symbols, d_h, d_pulse, d_out_ri, d_q_in, d_q_out, stream);
}

extern "C" int wf_mod_tile_geometry(int sps, int ntaps, int64_t nsym_total, int64_t *tile_len, int64_t *sym_per_tile,
                                    int64_t *ntiles_total)
{
    mod_params P;
    const bool ok = mod_setup(P, nsym_total, 1, ntaps, sps, 0.0);
    if (sps < 2 || sps > 256 || ntaps < 1 || nsym_total < 1) {
        wf_set_error("wf_mod_tile_geometry: sps %d ntaps %d nsym %lld outside 2 <= sps <= 256, ntaps >= 1, nsym >= 1",
                     sps, ntaps, (long long)nsym_total);
        if (tile_len) *tile_len = 0;
        if (sym_per_tile) *sym_per_tile = 0;
        if (ntiles_total) *ntiles_total = 0;
        return WF_ERR_VALUE;
    }
    if (tile_len) *tile_len = (int64_t)MOD_ROWS * P.rs;
    if (sym_per_tile) *sym_per_tile = P.spt;
    if (ntiles_total) *ntiles_total = P.ntiles;
    return ok ? 0 : 1;
}
