// wf_viz.hip — the numeric content of the reference's plotting helpers as DATA products on the
// device: Welch power spectral density (what Axes.psd computes for
// waveforms/viz/psd.py:36-41), phase-tree traces (waveforms/viz/tree.py:64-70) and eye-diagram
// traces (waveforms/viz/eye.py:40-55).  No plotting here: the host mirror returns arrays.
#include "wf_common.h"

#define VIZ_THREADS 256
#define VIZ_MAX_LOG2 12          // segments of up to 4096 samples live in LDS (64 KB)

// One workgroup per segment (grid-strided): window, radix-2 decimation-in-time FFT in LDS,
// |X|^2 accumulated per bin in registers; partial sums per workgroup, reduced in a fixed order by
// welch_reduce_kernel (no floating-point atomics: the result does not depend on scheduling).
__global__ __launch_bounds__(VIZ_THREADS) void welch_kernel(const double2 *__restrict__ x, int64_t n, int64_t nseg, int nfft,
                                                             int m, double scale, const double *__restrict__ window,
                                                             double *__restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) double2 s_buf[];
    double2 *s_tw = s_buf + nfft;
    const int t = threadIdx.x;
    for (int j = t; j < nfft / 2; j += VIZ_THREADS) {
        double sn, cs;
        sincospi(-2.0 * (double)j / (double)nfft, &sn, &cs);
        s_tw[j] = make_double2(cs, sn);
    }
    double acc[(1 << VIZ_MAX_LOG2) / VIZ_THREADS];
#pragma unroll
    for (int q = 0; q < (1 << VIZ_MAX_LOG2) / VIZ_THREADS; ++q) acc[q] = 0.0;
    for (int64_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        __syncthreads();
        for (int i = t; i < nfft; i += VIZ_THREADS) {
            const int64_t a = seg * nfft + i;
            double2 v = a < n ? x[a] : make_double2(0.0, 0.0);          // a short signal is zero-padded to one segment
            const double w = window[i];
            v = make_double2(v.x * scale * w, v.y * scale * w);
            s_buf[__brev((unsigned)i) >> (32 - m)] = v;
        }
        for (int s = 1; s <= m; ++s) {
            __syncthreads();
            const int half = 1 << (s - 1), tstep = nfft >> s;
            for (int b = t; b < nfft / 2; b += VIZ_THREADS) {
                const int pos = b & (half - 1);
                const int i0 = ((b >> (s - 1)) << s) + pos, i1 = i0 + half;
                const double2 w = s_tw[pos * tstep], a = s_buf[i0], c = s_buf[i1];
                const double2 cw = make_double2(fma(c.x, w.x, -c.y * w.y), fma(c.x, w.y, c.y * w.x));
                s_buf[i0] = make_double2(a.x + cw.x, a.y + cw.y);
                s_buf[i1] = make_double2(a.x - cw.x, a.y - cw.y);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (1 << VIZ_MAX_LOG2) / VIZ_THREADS; ++q) {
            const int i = t + q * VIZ_THREADS;
            if (i < nfft) acc[q] += fma(s_buf[i].x, s_buf[i].x, s_buf[i].y * s_buf[i].y);
        }
    }
#pragma unroll
    for (int q = 0; q < (1 << VIZ_MAX_LOG2) / VIZ_THREADS; ++q) {
        const int i = t + q * VIZ_THREADS;
        if (i < nfft) partial[(size_t)blockIdx.x * nfft + ((i + nfft / 2) & (nfft - 1))] = acc[q];   // fftshift
    }
}

__global__ void welch_reduce_kernel(const double *__restrict__ partial, int nblocks, int nfft, double inv_nseg, double inv_wsum2,
                                    double *__restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nfft) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * nfft + k];
    out[k] = s * inv_nseg * inv_wsum2;
}

// Welch PSD as matplotlib.mlab.psd computes it with the arguments of waveforms/viz/psd.py:36-41
// (detrend none, noverlap 0, two-sided, scale_by_freq False): mean over the len // nfft segments of
// |FFT(window * scale * x)|^2, divided by (sum |window|)^2, bins in fftshift order.
// d_window: nfft doubles (np.hanning(nfft) for the reference's call), wsum = sum |window|.
// d_scratch: wf_welch_scratch_doubles(n, nfft) doubles.
extern "C" int64_t wf_welch_scratch_doubles(int64_t n, int nfft)
{
    if (n < 0 || nfft < 2) return -1;
    int64_t nseg = n / nfft;
    if (nseg < 1) nseg = 1;
    return (nseg < 256 ? nseg : 256) * (int64_t)nfft;
}

extern "C" int wf_welch_psd_c128(wf_ctx *ctx, const double *d_x_ri, int64_t n, int nfft, double scale, const double *d_window,
                                 double wsum, double *d_scratch, double *d_pxx, void *stream)
{
    WF_REQUIRE(ctx && n >= 1, "wf_welch_psd_c128: bad argument");
    int m = 0;
    while ((1 << m) < nfft) ++m;
    WF_REQUIRE(nfft >= 2 && (1 << m) == nfft && m <= VIZ_MAX_LOG2, "wf_welch_psd_c128: nfft %d must be a power of two <= %d", nfft,
               1 << VIZ_MAX_LOG2);
    WF_REQUIRE(d_x_ri && d_window && d_scratch && d_pxx && wsum != 0.0, "wf_welch_psd_c128: NULL argument");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_x_ri) & 15) == 0, "wf_welch_psd_c128: signal must be 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    int64_t nseg = n / nfft;
    if (nseg < 1) nseg = 1;
    const int grid = (int)(nseg < 256 ? nseg : 256);
    const size_t lds = (size_t)(nfft + nfft / 2) * sizeof(double2);
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(welch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t s = wf_stream(stream);
    hipLaunchKernelGGL(welch_kernel, dim3(grid), dim3(VIZ_THREADS), lds, s, reinterpret_cast<const double2 *>(d_x_ri), n, nseg, nfft,
                       m, scale, d_window, d_scratch);
    WF_LAUNCH_CHECK();
    hipLaunchKernelGGL(welch_reduce_kernel, dim3((nfft + 255) / 256), dim3(256), 0, s, d_scratch, grid, nfft, 1.0 / (double)nseg,
                       1.0 / (wsum * wsum), d_pxx);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// numpy's `%` for doubles with a positive divisor
__device__ __forceinline__ double viz_pymod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0 && r < 0.0) r += b;
    return r;
}

// Phase-tree traces (waveforms/viz/tree.py:64-70): chunk c = samples [c*len, (c+1)*len), phase =
// np.angle, np.unwrap inside the chunk, minus `off` (or minus the chunk's first value when
// use_first != 0).  One thread per chunk (len = modulo * sps is a few tens of samples).
__global__ void phase_tree_kernel(const double2 *__restrict__ x, int64_t nchunks, int len, int use_first, double off,
                                  double *__restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const double2 *p = x + c * len;
    double *o = out + c * len;
    const double PI = 3.14159265358979323846, TWO_PI = 6.28318530717958647692;
    double prev = atan2(p[0].y, p[0].x), cum = 0.0;
    const double first = prev;
    o[0] = prev - (use_first ? first : off);
    for (int j = 1; j < len; ++j) {
        const double cur = atan2(p[j].y, p[j].x);
        const double dd = cur - prev;
        double ddmod = viz_pymod(dd + PI, TWO_PI) - PI;                 // np.unwrap, discont = pi, period = 2 pi
        if (ddmod == -PI && dd > 0.0) ddmod = PI;
        double corr = ddmod - dd;
        if (fabs(dd) < PI) corr = 0.0;
        cum += corr;
        o[j] = (cur + cum) - (use_first ? first : off);
        prev = cur;
    }
}

extern "C" int wf_phase_tree_f64(wf_ctx *ctx, const double *d_x_ri, int64_t n, int sps, int modulo, int use_first, double off,
                                 double *d_out, void *stream)
{
    WF_REQUIRE(ctx && n >= 0 && sps >= 1 && modulo >= 1, "wf_phase_tree_f64: bad argument");
    const int len = sps * modulo;
    const int64_t nchunks = n / len;
    if (nchunks == 0) return WF_OK;
    WF_REQUIRE(d_x_ri && d_out, "wf_phase_tree_f64: NULL device pointer");
    WF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(phase_tree_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, wf_stream(stream),
                       reinterpret_cast<const double2 *>(d_x_ri), nchunks, len, use_first, off, d_out);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// Eye-diagram traces (waveforms/viz/eye.py:40-55): trace i = samples [i*len, i*len + len] (len + 1
// points, neighbouring traces share an end point), real and imaginary parts in separate planes,
// and the time axis (time[start + j] - time[start]) + t_offset.
__global__ void eye_kernel(const double *__restrict__ time, const double2 *__restrict__ x, int64_t ntraces, int len, double t_offset,
                           double *__restrict__ t_out, double *__restrict__ re_out, double *__restrict__ im_out)
{
    const int64_t total = ntraces * (len + 1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const int64_t i = q / (len + 1);
        const int j = (int)(q - i * (len + 1));
        const int64_t a = i * len;
        t_out[q] = (time[a + j] - time[a]) + t_offset;
        re_out[q] = x[a + j].x;
        im_out[q] = x[a + j].y;
    }
}

extern "C" int wf_eye_traces_c128(wf_ctx *ctx, const double *d_time, const double *d_x_ri, int64_t n, int sps, int modulo,
                                  double t_offset, double *d_t_out, double *d_re_out, double *d_im_out, void *stream)
{
    WF_REQUIRE(ctx && n >= 0 && sps >= 1 && modulo >= 1, "wf_eye_traces_c128: bad argument");
    const int len = sps * modulo;
    const int64_t ntraces = n >= 1 ? (n - 1) / len : 0;                  // range((time.size - 1) // (sps * modulo))
    if (ntraces == 0) return WF_OK;
    WF_REQUIRE(d_time && d_x_ri && d_t_out && d_re_out && d_im_out, "wf_eye_traces_c128: NULL device pointer");
    WF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(eye_kernel, dim3(wf_grid_for(ntraces * (len + 1), 256, 4096)), dim3(256), 0, wf_stream(stream), d_time,
                       reinterpret_cast<const double2 *>(d_x_ri), ntraces, len, t_offset, d_t_out, d_re_out, d_im_out);
    WF_LAUNCH_CHECK();
    return WF_OK;
}
