// wf_fir.hip — K3: zero-stuffed upsample + frequency-pulse FIR ("same" convolution).
// Replaces  interpolated[sps:-1:sps] = symbols*h ; np.convolve(interp, pulse, "same")
// (reference waveforms/cpm/modulate.py:91-99).
//
// Polyphase form: output sample n only sees the <= J = ceil(M/sps) symbols whose
// impulses fall under the pulse, and always the taps g[r], g[r+sps], ... with
// r = (n + c) mod sps, c = the "same" centring offset.  A workgroup walks tiles of
// ROWS x RS consecutive samples, RS a multiple of lcm(2, sps), so a thread's two
// adjacent outputs keep the same tap phases for the whole launch: its 2 x J taps
// live in registers, the tile's symbol amplitudes (symbol * h, zero outside the
// burst — which is also what implements the edge truncation of "same") are staged
// once in LDS, and every store is 16 B per lane, coalesced.  HBM-bound:
// 1 B read + 8*sps B written per symbol.
#include "wf_common.h"

#define FIR_THREADS 256
#define FIR_ROWS 8

struct fir_params {
    int64_t nsym;
    int64_t out_len;
    int sps, ntaps, nh;
    int c;       // index of full-convolution sample that lands on out[0]
    int rs;      // samples per row (multiple of lcm(2, sps), <= 2*FIR_THREADS)
    int64_t ntiles;
};

template <int JMAX>
__global__ __launch_bounds__(FIR_THREADS) void fir_kernel(const int8_t *__restrict__ symbols,
                                                           const double *__restrict__ hvec,
                                                           const double *__restrict__ pulse,
                                                           double *__restrict__ out, fir_params P)
{
    // window of symbol amplitudes for one tile: ROWS*rs/sps + JMAX + 2 entries (dynamic LDS)
    extern __shared__ double s_amp[];
    const int t = threadIdx.x;
    const int sps = P.sps;
    const bool active = 2 * t < P.rs;
    const int tile_len = FIR_ROWS * P.rs;     // multiple of sps
    const int sym_per_row = P.rs / sps;
    const int cq = P.c / sps;

    // per-thread constants: phase of the two outputs and their taps
    const int q0 = (2 * t + P.c) / sps;
    const int r0 = (2 * t + P.c) - q0 * sps;
    const int wrap = (r0 + 1 == sps) ? 1 : 0;
    const int r1 = wrap ? 0 : r0 + 1;
    // shifted tap vectors over the shared window a[i] = amp[top0 + 1 - i], i = 0..JMAX
    double tap0[JMAX + 1], tap1[JMAX + 1];
#pragma unroll
    for (int i = 0; i <= JMAX; ++i) {
        const int j0 = i - 1;             // out0 uses symbol top0 - j0
        const int k0 = r0 + j0 * sps;
        tap0[i] = (j0 >= 0 && k0 < P.ntaps) ? pulse[k0] : 0.0;
        const int j1 = i - 1 + wrap;      // out1 uses symbol top1 - j1, top1 = top0 + wrap
        const int k1 = r1 + j1 * sps;
        tap1[i] = (j1 >= 0 && j1 < JMAX && k1 < P.ntaps) ? pulse[k1] : 0.0;
    }
    // local index (inside s_amp) of symbol top0+1 for row 0; window starts at
    // mp1 = tile_base/sps + cq - JMAX + 1  ->  local = mp1 - that
    const int l_top0p1 = (q0 - cq) + JMAX;  // +1 for "top0+1", -1 for window start offset

    const int win = FIR_ROWS * sym_per_row + JMAX + 2;
    for (int64_t tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
        const int64_t tile_base = tile * tile_len;
        const int64_t mp1_lo = tile_base / sps + cq - JMAX + 1;
        __syncthreads();
        for (int k = t; k < win; k += FIR_THREADS) {
            const int64_t mp1 = mp1_lo + k;
            double a = 0.0;
            if (mp1 >= 1 && mp1 <= P.nsym) {
                const int64_t m = mp1 - 1;
                a = (double)symbols[m] * hvec[P.nh == 1 ? 0 : (int)(m % P.nh)];
            }
            s_amp[k] = a;
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int u = 0; u < FIR_ROWS; ++u) {
                const int64_t n = tile_base + (int64_t)u * P.rs + 2 * t;
                const double *a = &s_amp[l_top0p1 + u * sym_per_row];
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int i = 0; i <= JMAX; ++i) {
                    const double v = a[-i];
                    acc0 = fma(tap0[i], v, acc0);
                    acc1 = fma(tap1[i], v, acc1);
                }
                if (n + 1 < P.out_len) {
                    // non-temporal: the phase scan that reads this next runs 5 % faster when these
                    // lines are not left dirty in L2 (this kernel itself 2 % slower)
                    wf_store16_nt(reinterpret_cast<double2 *>(out + n), make_double2(acc0, acc1));
                } else if (n < P.out_len) {
                    out[n] = acc0;
                }
            }
        }
    }
}

static int gcd_int(int a, int b) { return b ? gcd_int(b, a % b) : a; }

extern "C" int64_t wf_fir_out_len(int64_t nsym, int sps, int ntaps)
{
    const int64_t npts = (nsym + 1) * (int64_t)sps;
    return npts >= ntaps ? npts : ntaps;
}

extern "C" int wf_upsample_fir_f64(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym,
                                   const double *d_h, int nh, const double *d_pulse, int ntaps,
                                   int sps, double *d_out, void *stream)
{
    WF_REQUIRE(ctx && d_h && d_pulse && d_out, "wf_upsample_fir_f64: NULL argument");
    WF_REQUIRE(nsym >= 0 && (nsym == 0 || d_symbols), "wf_upsample_fir_f64: bad symbols");
    // sps = 1: the reference's interpolated[sps:-1:sps] has N-1 slots for N symbols and
    // numpy raises ValueError (waveforms/cpm/modulate.py:96) — same error here.
    WF_REQUIRE(sps >= 2 || nsym == 0, "could not broadcast input array from shape (%lld,) into shape (%lld,)",
               (long long)nsym, (long long)(nsym - 1));
    WF_REQUIRE(sps >= 1 && sps <= 256 && nh >= 1 && ntaps >= 1,
               "wf_upsample_fir_f64: sps %d nh %d ntaps %d", sps, nh, ntaps);
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_out) & 15) == 0, "wf_upsample_fir_f64: d_out alignment");
    const int J = (ntaps + sps - 1) / sps;
    WF_REQUIRE(J <= 33, "wf_upsample_fir_f64: pulse longer than 33 symbols (%d taps @ %d sps)", ntaps, sps);
    WF_HIP(hipSetDevice(ctx->device));
    fir_params P;
    P.nsym = nsym;
    const int64_t npts = (nsym + 1) * (int64_t)sps;
    P.out_len = npts >= ntaps ? npts : ntaps;
    // np.convolve swaps operands when the pulse is the longer one
    P.c = (int)(((npts >= ntaps ? (int64_t)ntaps : npts) - 1) / 2);
    P.sps = sps;
    P.ntaps = ntaps;
    P.nh = nh;
    const int l = sps / gcd_int(sps, 2) * 2;  // lcm(2, sps)
    P.rs = (2 * FIR_THREADS / l) * l;
    const int64_t tile_len = (int64_t)FIR_ROWS * P.rs;
    P.ntiles = (P.out_len + tile_len - 1) / tile_len;
    const int grid = (int)(P.ntiles < 2048 ? P.ntiles : 2048);
    hipStream_t s = wf_stream(stream);
#define FIR_LAUNCH(JM)                                                                          \
    hipLaunchKernelGGL(fir_kernel<JM>, dim3(grid), dim3(FIR_THREADS),                           \
                       (size_t)(FIR_ROWS * (P.rs / sps) + JM + 2) * sizeof(double), s, d_symbols, \
                       d_h, d_pulse, d_out, P)
    if (J <= 4) FIR_LAUNCH(4);
    else if (J <= 9) FIR_LAUNCH(9);
    else if (J <= 17) FIR_LAUNCH(17);
    else FIR_LAUNCH(33);
#undef FIR_LAUNCH
    WF_LAUNCH_CHECK();
    return WF_OK;
}
