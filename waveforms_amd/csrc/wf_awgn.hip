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

__global__ __launch_bounds__(256) void awgn_kernel(const double *in, int64_t n,
                                                    double rot_re, double rot_im, double sigma,
                                                    uint64_t seed, uint64_t stream_id,
                                                    uint64_t first_index, double *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        double nr, ni;
        wf_gaussian_pair(first_index + (uint64_t)k, stream_id, seed, sigma, &nr, &ni);
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
