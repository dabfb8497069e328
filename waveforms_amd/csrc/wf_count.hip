// wf_count.hip — K11: symbol / bit error counting.
// Replaces np.where(detected[:m] - reference[:m]) of the reference harness
// (examples/soqpsk_detection.py:200-209).  Integer path, exact.
#include "wf_common.h"

__global__ __launch_bounds__(256) void count_errors_kernel(const int8_t *__restrict__ det_syms,
                                                            const int8_t *__restrict__ ref_syms,
                                                            const uint8_t *__restrict__ det_bits,
                                                            const uint8_t *__restrict__ ref_bits, int64_t m,
                                                            unsigned long long *__restrict__ counts)
{
    __shared__ long long s_part[2][4];
    long long se = 0, be = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // 16 elements per thread and iteration; the arrays may start at any byte offset
    // (the detector output is compared from element `length` on), gfx950 global loads
    // need no alignment
    const int64_t nvec = m / 16;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        ulonglong2 a, b, c, d;
        __builtin_memcpy(&a, det_syms + 16 * v, 16);
        __builtin_memcpy(&b, ref_syms + 16 * v, 16);
        __builtin_memcpy(&c, det_bits + 16 * v, 16);
        __builtin_memcpy(&d, ref_bits + 16 * v, 16);
        // count non-zero bytes of the XOR: fold each byte to its low bit
        auto nz = [](unsigned long long x) {
            x |= x >> 4; x |= x >> 2; x |= x >> 1;
            return __popcll(x & 0x0101010101010101ull);
        };
        se += nz(a.x ^ b.x) + nz(a.y ^ b.y);
        be += nz(c.x ^ d.x) + nz(c.y ^ d.y);
    }
    for (int64_t k = 16 * nvec + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
        se += det_syms[k] != ref_syms[k];
        be += det_bits[k] != ref_bits[k];
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

extern "C" int wf_count_errors(wf_ctx *ctx, const int8_t *d_det_syms, const int8_t *d_ref_syms,
                               const uint8_t *d_det_bits, const uint8_t *d_ref_bits, int64_t m,
                               int64_t *d_counts, void *stream)
{
    WF_REQUIRE(ctx && m >= 0 && d_counts, "wf_count_errors: bad argument");
    if (m == 0) return WF_OK;
    WF_REQUIRE(d_det_syms && d_ref_syms && d_det_bits && d_ref_bits, "wf_count_errors: NULL device pointer");
    WF_HIP(hipSetDevice(ctx->device));
    const unsigned grid = wf_grid_for(m, 256 * 16 * 8, 256);   /* <= 512 same-address atomics */
    hipLaunchKernelGGL(count_errors_kernel, dim3(grid), dim3(256), 0, wf_stream(stream),
                       d_det_syms, d_ref_syms, d_det_bits, d_ref_bits, m,
                       reinterpret_cast<unsigned long long *>(d_counts));
    WF_LAUNCH_CHECK();
    return WF_OK;
}
