// wf_awgn.hip — K5: complex AWGN injection.
// Device counterpart of generate_complex_awgn (reference waveforms/noise.py:8-32) fused
// with the channel of the example:  r = s * exp(-j pi/4) + noise
// (reference examples/soqpsk_detection.py:85-89).
//
// The reference draws from numpy's PCG64 + ziggurat, a sequential generator with a
// data-dependent number of draws per sample; it cannot be evaluated in parallel, so
// the host API keeps honouring a caller-supplied numpy Generator (waveforms_amd/noise.py)
// and THIS kernel is the Monte-Carlo source: one Philox4x32-10 block per PAIR of complex
// samples (counter = absolute sample index >> 1, stream id; key = seed), two 32-bit
// uniforms per sample, Box-Muller in fp64 (Gaussian tail to 6.66 sigma).  Counter-based => any shard / chunk of a stream reproduces
// independently of launch geometry (what the 8-GPU BER sweep relies on).
// Element-wise, 16 B in + 16 B out per lane: HBM-bound (32 B/sample).
#include "wf_common.h"

__global__ __launch_bounds__(256) void awgn_kernel(const double *in, int64_t n, double rot_re, double rot_im,
                                                    double sigma, uint64_t seed, uint64_t stream_id,
                                                    uint64_t first_index, double *out)
{
    __shared__ double2 s_tab[256];   // log + sincos tables: LDS, because this loop stores (see wf_tabs_lds)
    wf_stage_tables<1, 0>(s_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const wf_tabs_lds<1, 0> tb{s_tab};
    // one thread per PAIR of absolute sample indices (2P, 2P+1): one Philox block serves both.
    // A wave covers 128 consecutive samples; they travel between HBM and the lanes through a
    // wave-private LDS strip so that every load / store instruction touches 64 consecutive
    // samples (1 KB) — a lane moving its own pair directly issues 16 B accesses at a 32 B stride,
    // which wrote 1.56 GB for a 1.28 GB burst (half-filled lines per instruction).
    __shared__ double2 s_xp[2 * 256];
    const int lane = threadIdx.x & 63;
    double2 *xw = s_xp + (threadIdx.x >> 6) * (2 * WF_WAVE);
    const uint64_t pair0 = first_index >> 1;
    const int64_t npairs = (int64_t)(((first_index + (uint64_t)n - 1) >> 1) - pair0) + 1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t q0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63);      // first pair of this wave
    const double2 *in2 = reinterpret_cast<const double2 *>(in);
    double2 *out2 = reinterpret_cast<double2 *>(out);
    for (int64_t qw = q0; qw < npairs; qw += stride) {
        const uint64_t pair = pair0 + (uint64_t)(qw + lane);
        const int64_t kb = (int64_t)(2 * (pair0 + (uint64_t)qw) - first_index);   // local index of the wave's first sample (may be -1)
        const int64_t ka = kb + lane, kc = ka + WF_WAVE;
        const bool oka = ka >= 0 && ka < n, okc = kc >= 0 && kc < n;
        double2 va = make_double2(0.0, 0.0), vc = va;
        if (in2) {
            if (oka) va = in2[ka];
            if (okc) vc = in2[kc];
        }
        xw[lane] = va;
        xw[WF_WAVE + lane] = vc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double2 v0 = xw[2 * lane], v1 = xw[2 * lane + 1];
        double g[4];
        wf_gaussian_two(pair, stream_id, seed, sigma, tb, g);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // every lane has read its pair before the strip is reused
        xw[2 * lane] = make_double2(fma(v0.x, rot_re, fma(-v0.y, rot_im, g[0])), fma(v0.x, rot_im, fma(v0.y, rot_re, g[1])));
        xw[2 * lane + 1] = make_double2(fma(v1.x, rot_re, fma(-v1.y, rot_im, g[2])), fma(v1.x, rot_im, fma(v1.y, rot_re, g[3])));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double2 ra = xw[lane], rc = xw[WF_WAVE + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (oka) wf_store16_nt(out2 + ka, ra);   // streamed once, read once by the next kernel
        if (okc) wf_store16_nt(out2 + kc, rc);
    }
}

extern "C" int wf_awgn_c128(wf_ctx *ctx, const double *d_in_ri, int64_t n, double rot_re,
                            double rot_im, double sigma, uint64_t seed, uint64_t stream_id,
                            uint64_t first_index, double *d_out_ri, void *stream)
{
    WF_REQUIRE(ctx && n >= 0, "wf_awgn_c128: bad argument");
    if (n == 0) return WF_OK;
    WF_REQUIRE(d_out_ri && (reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_in_ri) & 15) == 0,
               "wf_awgn_c128: device pointers must be 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(awgn_kernel, dim3(wf_grid_for((n + 1) / 2 + 1, 256, 256 * 16)), dim3(256), 0, wf_stream(stream),
                       d_in_ri, n, rot_re, rot_im, sigma, seed, stream_id, first_index, d_out_ri);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// Box-Muller on caller-supplied 32-bit words (2 per complex sample: radius word, angle word) —
// the transform of the Gaussian source without Philox in front, so that its edge cases
// (u1 = 2^-32, u1 = 1, u2 = 0, ...) can be driven directly.
__global__ void box_muller32_kernel(const uint32_t *__restrict__ words, int64_t n, double sigma, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        double re, im;
        wf_box_muller32(words[2 * k], words[2 * k + 1], sigma, wf_tabs_global{}, &re, &im);
        *reinterpret_cast<double2 *>(out + 2 * k) = make_double2(re, im);
    }
}

extern "C" int wf_box_muller32_c128(wf_ctx *ctx, const uint32_t *d_words, int64_t n, double sigma, double *d_out_ri,
                                    void *stream)
{
    WF_REQUIRE(ctx && n >= 0, "wf_box_muller32_c128: bad argument");
    if (n == 0) return WF_OK;
    WF_REQUIRE(d_words && d_out_ri && (reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0, "wf_box_muller32_c128: bad device pointer");
    WF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(box_muller32_kernel, dim3(wf_grid_for(n, 256, 4096)), dim3(256), 0, wf_stream(stream), d_words, n,
                       sigma, d_out_ri);
    WF_LAUNCH_CHECK();
    return WF_OK;
}
