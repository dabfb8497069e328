// wf_phase.hip — K4: phase accumulation (mod sps) + complex exponential.
// Replaces the per-sample Python loop of frequency_modulate
// (reference waveforms/cpm/modulate.py:28-54):
//     revs = (revs + sample) % sps ; phase = revs*2pi/sps + phi0 ; out = exp(1j*phase)
// and phase_modulate (modulate.py:12-25).
//
// Single-pass chained prefix scan ("decoupled look-back"): the input is read once
// and the output written once (8 B + 16 B per sample — the HBM roofline of this
// stage).  Tiles of 4096 samples are claimed through an atomic ticket, so every
// predecessor of a tile is already running when it starts.  Only the accumulated
// phase MODULO sps matters, so a tile's aggregate is published as a 62-bit
// fixed-point fraction of a revolution with a 2-bit status in ONE 64-bit word
// (relaxed agent-scope atomic = one `sc1` 8-byte store; no separate flag, no
// fence, and modular addition of the fixed-point words is exact and
// order-independent).  Inside a tile sums are fp64 (fp32 fails the 1e-6 parity
// bound); row scans use wave shuffles, cross-wave totals go through LDS.
#include "wf_common.h"

#define PH_THREADS 256
#ifndef PH_ROWS
#define PH_ROWS 8
#endif
#define PH_ROW (2 * PH_THREADS)
#define PH_TILE (PH_ROWS * PH_ROW)
#define PH_WAVES (PH_THREADS / WF_WAVE)
#define PH_DESC0 8  // descriptors start at scan[PH_DESC0]

#define PH_MASK ((1ull << 62) - 1)
#define PH_FLAG_A (1ull << 62)
#define PH_FLAG_P (2ull << 62)
#define PH_SPIN_LIMIT (1 << 22)

struct phase_params {
    int64_t n;
    int64_t ntiles;
    double sps;      // modulus as a double
    double inv_sps;
    double phi0_turns;  // phi0 / (2 pi)
    uint64_t q_in;   // carried-in revs as a 62-bit fraction of sps
};

__device__ __forceinline__ double mod_pos(double v, double sps, double inv_sps)
{
    // v mod sps in [0, sps): k*sps is exact (sps integer, |k| small), the fma exact.
    double k = floor(v * inv_sps);
    double m = fma(-k, sps, v);
    if (m < 0.0) m += sps;
    if (m >= sps) m -= sps;
    return m;
}

__device__ __forceinline__ uint64_t to_fixed(double revs_mod, double inv_sps)
{
    // revs_mod in [0, sps] -> fraction of a revolution in 62 bits
    return (uint64_t)(revs_mod * inv_sps * 0x1.0p62) & PH_MASK;
}

__device__ __forceinline__ double from_fixed(uint64_t q, double sps)
{
    return (double)q * 0x1.0p-62 * sps;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int d = WF_WAVE / 2; d >= 1; d >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, d, WF_WAVE);
    return v;
}

__global__ __launch_bounds__(PH_THREADS) void phase_kernel(const double *__restrict__ f,
                                                            double *__restrict__ out,
                                                            uint64_t *__restrict__ scan, phase_params P,
                                                            unsigned *__restrict__ fault,
                                                            double *__restrict__ revs_out)
{
    __shared__ double s_tot[PH_ROWS * PH_WAVES];
    __shared__ long long s_tile;
    __shared__ uint64_t s_prefix;
    __shared__ double2 s_cis[128];   // sincos sector table in LDS (persistent store loop: see wf_sincos_sectors)
    __shared__ double2 s_xp[2 * PH_THREADS];   // wave-private strips for the store transpose
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    uint64_t *desc = scan + PH_DESC0;
    wf_stage_cis_table(s_cis, t, PH_THREADS);   // the tile loop opens with a barrier

    for (long long iter_ = 0;; ++iter_) {
        (void)iter_;
        wf_lds_barrier();  // previous iteration's LDS reads are done
#ifdef WF_ABL_NO_TICKET    // ablation only: static tile order (can deadlock in general)
        if (t == 0) s_tile = (long long)blockIdx.x + (long long)gridDim.x * iter_;
#else
        if (t == 0) s_tile = (long long)atomicAdd((unsigned long long *)&scan[0], 1ull);
#endif
        wf_lds_barrier();
        const int64_t tile = s_tile;
        if (tile >= P.ntiles) break;
        const int64_t base = tile * PH_TILE;

        double x0[PH_ROWS], x1[PH_ROWS], ex[PH_ROWS];
#pragma unroll
        for (int u = 0; u < PH_ROWS; ++u) {
            const int64_t i = base + u * PH_ROW + 2 * t;
            if (i + 1 < P.n) {
                const double2 v = *reinterpret_cast<const double2 *>(f + i);
                x0[u] = v.x;
                x1[u] = v.y;
            } else {
                x0[u] = i < P.n ? f[i] : 0.0;
                x1[u] = 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < PH_ROWS; ++u) {
            const double inc = wf_wave_incl_scan(x0[u] + x1[u]);
            ex[u] = wf_wave_shr1(inc);
            if (lane == 63) s_tot[u * PH_WAVES + wave] = inc;
        }
        wf_lds_barrier();
        double off[PH_ROWS];
        double running = 0.0;
#pragma unroll
        for (int u = 0; u < PH_ROWS; ++u) {
#pragma unroll
            for (int w = 0; w < PH_WAVES; ++w) {
                if (w == wave) off[u] = running;
                running += s_tot[u * PH_WAVES + w];
            }
        }
        // `running` = tile total, identical in every thread
        const uint64_t q_agg = to_fixed(mod_pos(running, P.sps, P.inv_sps), P.inv_sps);

#ifdef WF_ABL_NO_LOOKBACK   // ablation only: wrong prefixes
        if (t == 0) s_prefix = 0;
        if (false) {
#else
        if (wave == 0) {
#endif
            if (lane == 0)
                __hip_atomic_store(&desc[tile], PH_FLAG_A | q_agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t q_ex = 0;
            int64_t look = tile - 1;
            int spins = 0;
            bool done = false;
            while (!done) {
                const int64_t idx = look - lane;
                uint64_t d;
                for (;;) {
                    if (idx >= 0)
                        d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        d = PH_FLAG_P | (idx == -1 ? P.q_in : 0ull);
                    const unsigned long long ready = __ballot((d >> 62) != 0);
                    const unsigned long long pm = __ballot((d >> 62) == 2);
                    // lanes nearer than (and including) the nearest inclusive prefix must be ready
                    const unsigned long long need = pm ? ((2ull << __builtin_ctzll(pm)) - 1ull) : ~0ull;
                    if ((ready & need) == need) break;
                    if (++spins > PH_SPIN_LIMIT) {  // never hang the GPU: flag and fall through
                        if (lane == 0) atomicOr(fault, WF_FAULT_SCAN_TIMEOUT);
                        if ((d >> 62) == 0) d = PH_FLAG_P;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                const unsigned long long pm = __ballot((d >> 62) == 2);
                uint64_t contrib = d & PH_MASK;
                if (pm) {
                    if (lane > __builtin_ctzll(pm)) contrib = 0;
                    done = true;
                }
                q_ex += wave_sum_u64(contrib);
                look -= WF_WAVE;
            }
            q_ex &= PH_MASK;
            if (lane == 0) {
                s_prefix = q_ex;
                const uint64_t q_inc = (q_ex + q_agg) & PH_MASK;
                __hip_atomic_store(&desc[tile], PH_FLAG_P | q_inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (revs_out && tile == P.ntiles - 1) *revs_out = from_fixed(q_inc, P.sps);
            }
        }
        wf_lds_barrier();
        const double prefix = from_fixed(s_prefix, P.sps);
#pragma unroll
        for (int u = 0; u < PH_ROWS; ++u) {
            const int64_t i = base + u * PH_ROW + 2 * t;
            const double v0 = prefix + (off[u] + ex[u] + x0[u]);
            const double v1 = v0 + x1[u];
            const double r0 = mod_pos(v0, P.sps, P.inv_sps);
            const double r1 = mod_pos(v1, P.sps, P.inv_sps);
            // phase = revs * 2 pi / sps + phi0, evaluated in turns (exact quadrant reduction)
            double s0, c0, s1, c1;
#ifdef WF_ABL_NO_SINCOS
            s0 = r0; c0 = r0 + 1; s1 = r1; c1 = r1 + 1;
#else
            wf_sincos_sectors(s_cis, fma(r0, 128.0 * P.inv_sps, 128.0 * P.phi0_turns), &s0, &c0);
            wf_sincos_sectors(s_cis, fma(r1, 128.0 * P.inv_sps, 128.0 * P.phi0_turns), &s1, &c1);
#endif
            // wave-private transpose (as in wf_modulate.hip): a lane holding two adjacent samples
            // would store 16 B at a 32 B stride; this way every store instruction writes 64
            // consecutive samples
            (void)i;
            double2 *xw = s_xp + wave * (2 * WF_WAVE);
            xw[2 * lane] = make_double2(c0, s0);
            xw[2 * lane + 1] = make_double2(c1, s1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double2 xa = xw[lane], xb = xw[WF_WAVE + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();   // strip read before the next row overwrites it
            const int64_t na = base + u * PH_ROW + wave * (2 * WF_WAVE) + lane, nb = na + WF_WAVE;
            double2 *o = reinterpret_cast<double2 *>(out);
            if (na < P.n) wf_store16_nt(o + na, xa);   // streamed once, read once by the next kernel
            if (nb < P.n) wf_store16_nt(o + nb, xb);
        }
    }
}

extern "C" int wf_phase_cexp_f64(wf_ctx *ctx, const double *d_freq, int64_t n, int sps, double phi0,
                                 double revs_in, double *d_out_ri, double *d_revs_out, void *stream)
{
    WF_REQUIRE(ctx != nullptr, "wf_phase_cexp_f64: ctx is NULL");
    WF_REQUIRE(n >= 0 && sps >= 1 && sps <= 4096, "wf_phase_cexp_f64: n %lld sps %d", (long long)n, sps);
    WF_HIP(hipSetDevice(ctx->device));
    hipStream_t s = wf_stream(stream);
    if (n == 0) {
        if (d_revs_out) WF_HIP(hipMemcpyAsync(d_revs_out, &revs_in, sizeof(double), hipMemcpyHostToDevice, s));
        return WF_OK;
    }
    WF_REQUIRE(d_freq && d_out_ri && (reinterpret_cast<uintptr_t>(d_freq) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0,
               "wf_phase_cexp_f64: device pointers must be non-NULL and 16-byte aligned");
    phase_params P;
    P.n = n;
    P.ntiles = (n + PH_TILE - 1) / PH_TILE;
    P.sps = (double)sps;
    P.inv_sps = 1.0 / (double)sps;
    P.phi0_turns = phi0 / (2.0 * M_PI);
    double r = fmod(revs_in, (double)sps);
    if (r < 0) r += sps;
    P.q_in = (uint64_t)(r / sps * 0x1.0p62) & PH_MASK;
    int rc = wf_ctx_reserve_scan(ctx, (size_t)P.ntiles + PH_DESC0);
    if (rc) return rc;
    WF_HIP(hipMemsetAsync(ctx->d_scan, 0, ((size_t)P.ntiles + PH_DESC0) * sizeof(uint64_t), s));
    const int grid = (int)(P.ntiles < 2048 ? P.ntiles : 2048);
    hipLaunchKernelGGL(phase_kernel, dim3(grid), dim3(PH_THREADS), 0, s, d_freq, d_out_ri, ctx->d_scan,
                       P, ctx->d_fault, d_revs_out);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

__global__ void phase_modulate_kernel(const double *__restrict__ phase, int64_t n, double sens,
                                      double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        double s, c;
        sincos(sens * phase[k], &s, &c);
        *reinterpret_cast<double2 *>(out + 2 * k) = make_double2(c, s);
    }
}

extern "C" int wf_phase_modulate_f64(wf_ctx *ctx, const double *d_phase, int64_t n, double sens,
                                     double *d_out_ri, void *stream)
{
    WF_REQUIRE(ctx && n >= 0, "wf_phase_modulate_f64: bad argument");
    if (n == 0) return WF_OK;
    WF_REQUIRE(d_phase && d_out_ri && (reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0,
               "wf_phase_modulate_f64: bad device pointer");
    WF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(phase_modulate_kernel, dim3(wf_grid_for(n, 256, 4096)), dim3(256), 0,
                       wf_stream(stream), d_phase, n, sens, d_out_ri);
    WF_LAUNCH_CHECK();
    return WF_OK;
}
