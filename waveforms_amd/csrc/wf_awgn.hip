// wf_awgn.hip — K5: complex AWGN injection.
// Device counterpart of generate_complex_awgn (reference waveforms/noise.py:8-32) fused
// with the channel of the example:  r = s * exp(-j pi/4) + noise
// (reference examples/soqpsk_detection.py:85-89).
//
// The reference draws from numpy's PCG64 + ziggurat, a sequential generator with a
// data-dependent number of draws per sample; it cannot be evaluated in parallel, so
// the host API keeps honouring a caller-supplied numpy Generator (waveforms_amd/noise.py)
// and THIS kernel is the Monte-Carlo source: one Philox4x32-10 block per complex
// sample (counter = sample index, stream id; key = seed), two 53-bit uniforms,
// Box-Muller in fp64.  Counter-based => any shard / chunk of a stream reproduces
// independently of launch geometry (what the 8-GPU BER sweep relies on).
// Element-wise, 16 B in + 16 B out per lane: HBM-bound (32 B/sample).
#include "wf_common.h"

struct philox_out {
    uint32_t x0, x1, x2, x3;
};

__device__ __forceinline__ philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

__device__ __forceinline__ void gaussian_pair(uint64_t idx, uint64_t stream_id, uint64_t seed,
                                              double sigma, double *re, double *im)
{
#ifdef WF_ABL_NO_PHILOX   // ablation only: NOT a valid generator
    const philox_out p = {(uint32_t)idx * 2654435761u, (uint32_t)(idx >> 7) ^ (uint32_t)seed,
                          (uint32_t)idx * 40503u, (uint32_t)stream_id ^ (uint32_t)idx};
#else
    const philox_out p = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream_id,
                                       (uint32_t)(stream_id >> 32), (uint32_t)seed,
                                       (uint32_t)(seed >> 32));
#endif
    const uint64_t a = ((uint64_t)p.x1 << 32) | p.x0;
    const uint64_t b = ((uint64_t)p.x3 << 32) | p.x2;
    const double u1 = (double)((a >> 11) + 1) * 0x1.0p-53;  // (0, 1]
    const double u2 = (double)(b >> 11) * 0x1.0p-53;        // [0, 1)
#ifdef WF_ABL_NO_LOG
    const double r = sigma * u1;
#else
    const double r = sigma * wf_sqrt_pos(-2.0 * wf_log_normal(u1));
#endif
    double s, c;
#ifdef WF_ABL_NO_SINCOS
    s = u2; c = 1.0 - u2;
#else
    wf_sincos_turns(u2, &s, &c);
#endif
    *re = r * c;
    *im = r * s;
}

__global__ __launch_bounds__(256) void awgn_kernel(const double *in, int64_t n,
                                                    double rot_re, double rot_im, double sigma,
                                                    uint64_t seed, uint64_t stream_id,
                                                    uint64_t first_index, double *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        double nr, ni;
        gaussian_pair(first_index + (uint64_t)k, stream_id, seed, sigma, &nr, &ni);
        double re = nr, im = ni;
        if (in) {
            const double2 v = *reinterpret_cast<const double2 *>(in + 2 * k);
            re = fma(v.x, rot_re, fma(-v.y, rot_im, nr));
            im = fma(v.x, rot_im, fma(v.y, rot_re, ni));
        }
        *reinterpret_cast<double2 *>(out + 2 * k) = make_double2(re, im);
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
    hipLaunchKernelGGL(awgn_kernel, dim3(wf_grid_for(n, 256, 256 * 16)), dim3(256), 0, wf_stream(stream),
                       d_in_ri, n, rot_re, rot_im, sigma, seed, stream_id, first_index, d_out_ri);
    WF_LAUNCH_CHECK();
    return WF_OK;
}
