// wf_mfbank.hip — K6/K7: decimating matched-filter correlation bank.
// Replaces the 3 x np.convolve(received, taps_alpha, "same") of the pulse-truncation
// bank (reference examples/soqpsk_detection.py:141-156) and the 6 convolutions of the
// PAM bank (:164-173; the conj(pseudo-symbol) weights are folded into 3 complex tap
// rows on the host), evaluated ONLY at the samples the detector consumes (:189-196):
//     out[k][f] = sum_t r[first + k*step + c - t] * taps[f][t],   c = (ntaps-1)/2.
// With step = 1 it is the plain full-rate bank.
//
// A workgroup stages the contiguous input span of its outputs in LDS with coalesced
// 16 B loads (each received sample is read from HBM once), then every thread owns one
// output column and runs the taps out of LDS.  For an even `step` one 16 B pad slot per
// `step` samples makes the ds_read_b128 column accesses bank-conflict free; taps are
// wave-uniform (broadcast LDS reads in the 3 x 9 fast path, scalar loads otherwise).
// 128 B read + nfilt*16 B written per symbol at sps = 8: HBM-bound for the 9-tap PT bank.
// NOISE instantiations apply the channel (derotation + Philox AWGN, wf_awgn.hip) while
// staging — then the kernel is bound by the vector pipe — and PACK writes the 32 B
// detector-packed row instead of 3 complex outputs.
#include "wf_common.h"

#define MF_THREADS 256
#define MF_LDS_SLOTS 3072  // 48 KiB of complex128
#define MF_MAXI 12         // staged samples per thread: MF_LDS_SLOTS / MF_THREADS

struct mf_params {
    int64_t nsamp, first, ncols;
    int step, ntaps, nfilt;
    int c;
    int ob;    // outputs per workgroup iteration (<= MF_THREADS)
    int span;  // input samples staged per iteration
    int pad;   // 1: one pad slot every `step` samples
    int dump;  // LDS slot that absorbs the partial pairs of the interior staging path
    int taps_slot;   // step-8, any-length bank: LDS slot of the tap copy [NF][ntaps] (rows >= nfilt zero)
    int64_t nblk;
    // fused channel (NOISE instantiation): staged sample = r*rot + sigma*N(idx)
    double rot_re, rot_im, sigma;
    uint64_t seed, stream_id, first_index;
    const uint64_t *dyn_index;   // optional device addend to first_index (graph replay of a stream)
    int pack_par0;               // PACK instantiation: parity of the detector call index of column 0
};

// PACK (3-filter bank only): write the detector-packed row {Re z1, Im z1, a, b} — the 4 of the 6
// real components the 4-state detector reads from a row (wf_viterbi.hip: (a, b) = (Re z0, Im z2)
// on even calls, (Im z0, Re z2) on odd ones) — 32 B per symbol instead of 48.
template <int NF, bool NOISE, int STEP, int NTAPS, bool PACK = false>   // STEP / NTAPS: compile-time fast path (0 = take P.step / P.ntaps)
__global__ __launch_bounds__(MF_THREADS) void mf_bank_kernel(const double *__restrict__ r,
                                                              const double *__restrict__ taps,
                                                              double *__restrict__ out, mf_params P)
{
    extern __shared__ double2 s_win[];
    const int t = threadIdx.x;
    const int step = STEP ? STEP : P.step;
    const int pad = STEP ? (STEP % 2 == 0) : P.pad;
    const int ntaps = NTAPS ? NTAPS : P.ntaps;
    // fast path: the taps live in LDS (broadcast ds_read_b128) — as scalar operands the 54 doubles
    // of the 3 x 9 bank overflow the SGPR file and spill through v_writelane/v_readlane
    __shared__ double2 s_taps[NTAPS ? NF * NTAPS : 1];
    if (NTAPS && STEP && t < NF * NTAPS) s_taps[t] = reinterpret_cast<const double2 *>(taps)[t];
    // (read straight from global memory, block-uniform: no barrier needed before the test)
    bool sym_taps = NTAPS && STEP && NF == 3;
    if (sym_taps) {
        const double2 *tg = reinterpret_cast<const double2 *>(taps);
        for (int j = 0; j < (NTAPS ? NTAPS : 1); ++j) {
            const double2 t0 = tg[j], t1 = tg[NTAPS + j], t2 = tg[2 * NTAPS + j];
            sym_taps = sym_taps && t1.x == 1.0 && t1.y == 0.0 && t2.x == t0.x && t2.y == -t0.y;
        }
    }
    // The fused channel of the step-8 fast path keeps the Gaussian source's two tables in the
    // window's 256 PAD slots (slot 9 g + 8 is never written by the staging): LDS for free, and no
    // global table loads queued behind the previous iteration's stores on vmcnt.
    // Long banks at step 8 (the 73-tap PAM bank): the taps are copied behind the window once per
    // workgroup and read back as broadcasts; as scalar loads inside the tap loop every trip waited
    // for three dependent s_loads (fused channel + PAM bank: 1.12 ms).
    constexpr bool LONG8 = STEP == 8 && NTAPS == 0;
    if (LONG8) {
        double2 *lt = s_win + P.taps_slot;
        for (int k = t; k < NF * ntaps; k += MF_THREADS)
            lt[k] = k < P.nfilt * ntaps ? reinterpret_cast<const double2 *>(taps)[k] : make_double2(0.0, 0.0);
    }
    constexpr bool LDS_TABS = NOISE && STEP == 8;
    if (LDS_TABS) wf_stage_tables<9, 8>(s_win, t, MF_THREADS);   // first use is behind the loop's barrier
    for (int64_t blk = blockIdx.x; blk < P.nblk; blk += gridDim.x) {
        const int64_t k0 = blk * P.ob;
        // first input sample of the span: oldest sample of output k0
        const int64_t ws = P.first + k0 * step + P.c - (ntaps - 1);
        __syncthreads();
        if (NOISE) {
            // rolled loop over PAIRS of absolute sample indices (one Philox block per pair; the
            // Gaussian source is ~400 instructions); the next pair's loads are in flight while
            // the channel is applied to the current one.  All per-lane index math is 32-bit
            // window offsets; the burst bounds become a wave-uniform offset range [vlo, vhi).
            const int64_t a0 = (int64_t)(P.first_index + (P.dyn_index ? *P.dyn_index : 0ull)) + ws;   // absolute index of the window start
            const int odd = (int)(a0 & 1);                            // window starts on the odd half of a pair
            const uint64_t pair_lo = (uint64_t)(a0 >> 1);             // floor, also for negative a0
            const int64_t lo64 = -ws, hi64 = P.nsamp - ws;            // offsets of burst samples 0 and nsamp
            const int vlo = lo64 > 0 ? (lo64 < P.span ? (int)lo64 : P.span) : 0;
            const int vhi = hi64 < P.span ? (hi64 > 0 ? (int)hi64 : 0) : P.span;
            const double2 *rp = reinterpret_cast<const double2 *>(r) + ws;   // dereferenced only inside [vlo, vhi)
            if (vlo == 0 && vhi == P.span) {
                // interior window (all but the first and last workgroup iterations of a burst): every
                // offset is a burst sample, so there are no per-lane bounds.  The trip count is
                // block-uniform; the two partial pairs at the window ends load from a clamped address
                // and park their result in the dump slot behind the window.
                const int last = P.span - 1;
                const int dump = P.dump;
                const int niter = (P.span + 1 + 2 * MF_THREADS - 1) / (2 * MF_THREADS);
                int w0 = 2 * t - odd;
                auto ld = [&](int w) { return rp[min(max(w, 0), last)]; };
                double2 c0 = ld(w0), c1 = ld(w0 + 1);
                uint64_t pair = pair_lo + (uint64_t)t;                // (w0 + odd) >> 1 == t + it * MF_THREADS
#pragma unroll 1
                for (int it = 0; it < niter; ++it) {
                    const double2 n0 = ld(w0 + 2 * MF_THREADS), n1 = ld(w0 + 2 * MF_THREADS + 1);
                    double g[4];
                    if (LDS_TABS) wf_gaussian_two(pair, P.stream_id, wf_opaque_seed(P.seed), P.sigma, wf_tabs_lds<9, 8>{s_win}, g);
                    else wf_gaussian_two(pair, P.stream_id, wf_opaque_seed(P.seed), P.sigma, wf_tabs_global{}, g);
                    const int w1 = w0 + 1;
                    const int q0 = (w0 >= 0 && w0 <= last) ? w0 + (pad ? w0 / step : 0) : dump;
                    const int q1 = (w1 <= last) ? w1 + (pad ? w1 / step : 0) : dump;
                    s_win[q0] = make_double2(fma(c0.x, P.rot_re, fma(-c0.y, P.rot_im, g[0])),
                                             fma(c0.x, P.rot_im, fma(c0.y, P.rot_re, g[1])));
                    s_win[q1] = make_double2(fma(c1.x, P.rot_re, fma(-c1.y, P.rot_im, g[2])),
                                             fma(c1.x, P.rot_im, fma(c1.y, P.rot_re, g[3])));
                    c0 = n0;
                    c1 = n1;
                    w0 += 2 * MF_THREADS;
                    pair += MF_THREADS;
                }
            } else
            {
            auto fetch = [&](int w) {
                return (w >= vlo && w < vhi) ? rp[w] : make_double2(0.0, 0.0);
            };
            int w0 = 2 * t - odd;                                      // offset of the even half (-1 possible)
            double2 c0 = fetch(w0), c1 = fetch(w0 + 1);
#pragma unroll 1
            for (; w0 < P.span; w0 += 2 * MF_THREADS) {
                const double2 n0 = fetch(w0 + 2 * MF_THREADS), n1 = fetch(w0 + 2 * MF_THREADS + 1);
                const int w1 = w0 + 1;
                const bool in0 = w0 >= vlo && w0 < vhi, in1 = w1 >= vlo && w1 < vhi;
                double g[4] = {0.0, 0.0, 0.0, 0.0};
                if (in0 || in1)   // the channel of wf_awgn_c128, on the fly
                {
                    const uint64_t pr = pair_lo + (uint64_t)((w0 + odd) >> 1);
                    if (LDS_TABS) wf_gaussian_two(pr, P.stream_id, wf_opaque_seed(P.seed), P.sigma, wf_tabs_lds<9, 8>{s_win}, g);
                    else wf_gaussian_two(pr, P.stream_id, wf_opaque_seed(P.seed), P.sigma, wf_tabs_global{}, g);
                }
                if (w0 >= 0) {
                    const double2 x = in0 ? make_double2(fma(c0.x, P.rot_re, fma(-c0.y, P.rot_im, g[0])),
                                                         fma(c0.x, P.rot_im, fma(c0.y, P.rot_re, g[1])))
                                          : make_double2(0.0, 0.0);
                    s_win[w0 + (pad ? w0 / step : 0)] = x;
                }
                if (w1 < P.span) {
                    const double2 x = in1 ? make_double2(fma(c1.x, P.rot_re, fma(-c1.y, P.rot_im, g[2])),
                                                         fma(c1.x, P.rot_im, fma(c1.y, P.rot_re, g[3])))
                                          : make_double2(0.0, 0.0);
                    s_win[w1 + (pad ? w1 / step : 0)] = x;
                }
                c0 = n0;
                c1 = n1;
            }
            }
        } else {
            // all of this thread's loads are issued before the first one is consumed
            // (MF_MAXI x 16 B in flight per lane), then the LDS writes
            double2 v[MF_MAXI];
#pragma unroll
            for (int u = 0; u < MF_MAXI; ++u) {
                const int i = t + u * MF_THREADS;
                const int64_t s = ws + i;
                v[u] = make_double2(0.0, 0.0);
                if (i < P.span && s >= 0 && s < P.nsamp) v[u] = *reinterpret_cast<const double2 *>(r + 2 * s);
            }
            int q = t / step, rem = t - q * step;
            const int dq = MF_THREADS / step, dr = MF_THREADS - dq * step;
#pragma unroll
            for (int u = 0; u < MF_MAXI; ++u) {
                const int i = t + u * MF_THREADS;
                if (i < P.span) s_win[i + (pad ? q : 0)] = v[u];
                q += dq;
                rem += dr;
                if (rem >= step) {
                    rem -= step;
                    ++q;
                }
            }
        }
        __syncthreads();
        const int64_t k = k0 + t;
        if (t < P.ob && k < P.ncols) {
            double ar[NF], ai[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) ar[f] = ai[f] = 0.0;
            // window offset j = 0 is the OLDEST sample => tap index ntaps-1-j
            const int base = t * (step + pad);
            if (NTAPS && STEP) {
                // fully unrolled: LDS offsets and tap addresses are compile-time constants
                // (this instantiation is dispatched only for nfilt == NF)
                double2 x[NTAPS ? NTAPS : 1];
#pragma unroll
                for (int j = 0; j < NTAPS; ++j) x[j] = s_win[base + j + (STEP % 2 == 0 ? j / STEP : 0)];
                if (NF == 3 && sym_taps) {
                    // Pulse-truncation bank: filter 1 is all ones and filter 2 = conj(filter 0) (checked
                    // on the taps themselves, below the kernel's tap staging).  z1 is then a plain sum
                    // — bit-identical to the products with 1 + 0j — and z0, z2 share their four real
                    // sums: 36 fma + 22 add instead of 108 fma.
                    double A = 0.0, B = 0.0, C = 0.0, D = 0.0;
#pragma unroll
                    for (int j = 0; j < NTAPS; ++j) {
                        const double2 tp = s_taps[NTAPS - 1 - j];
                        A = fma(x[j].x, tp.x, A);
                        B = fma(x[j].y, tp.y, B);
                        C = fma(x[j].x, tp.y, C);
                        D = fma(x[j].y, tp.x, D);
                        ar[1] += x[j].x;
                        ai[1] += x[j].y;
                    }
                    ar[0] = A - B;
                    ai[0] = C + D;
                    ar[NF - 1] = A + B;
                    ai[NF - 1] = D - C;
                } else
#pragma unroll
                for (int f = 0; f < NF; ++f) {
#pragma unroll
                    for (int j = 0; j < NTAPS; ++j) {
                        const double2 tp = s_taps[f * NTAPS + (NTAPS - 1 - j)];
                        ar[f] = fma(x[j].x, tp.x, fma(-x[j].y, tp.y, ar[f]));
                        ai[f] = fma(x[j].x, tp.y, fma(x[j].y, tp.x, ai[f]));
                    }
                }
            } else if (LONG8) {
                const double2 *lt = s_win + P.taps_slot;
                const int ng = ntaps >> 3;
                for (int g = 0; g < ng; ++g) {
                    const double2 *xw = s_win + base + 9 * g;          // 8 samples, then one pad slot
                    const double2 *tg = lt + (ntaps - 1 - 8 * g);      // tap of sample 8 g, going down
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const double2 x = xw[r];
#pragma unroll
                        for (int f = 0; f < NF; ++f) {
                            const double2 tp = tg[f * ntaps - r];
                            ar[f] = fma(x.x, tp.x, fma(-x.y, tp.y, ar[f]));
                            ai[f] = fma(x.x, tp.y, fma(x.y, tp.x, ai[f]));
                        }
                    }
                }
                for (int j = 8 * ng; j < ntaps; ++j) {                 // ntaps % 8 trailing taps
                    const double2 x = s_win[base + j + (j >> 3)];
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        const double2 tp = lt[f * ntaps + (ntaps - 1 - j)];
                        ar[f] = fma(x.x, tp.x, fma(-x.y, tp.y, ar[f]));
                        ai[f] = fma(x.x, tp.y, fma(x.y, tp.x, ai[f]));
                    }
                }
            } else {
                int extra = 0, jm = 0;
                for (int j = 0; j < ntaps; ++j) {
                    const double2 x = s_win[base + j + extra];
                    const int tt = ntaps - 1 - j;
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if (f < P.nfilt) {
                            const double tr = taps[2 * (f * ntaps + tt)];
                            const double ti = taps[2 * (f * ntaps + tt) + 1];
                            ar[f] = fma(x.x, tr, fma(-x.y, ti, ar[f]));
                            ai[f] = fma(x.x, ti, fma(x.y, tr, ai[f]));
                        }
                    }
                    if (++jm == step) {
                        jm = 0;
                        extra += pad;
                    }
                }
            }
            if (PACK) {
                const bool odd_call = ((P.pack_par0 + (int)(k & 1)) & 1) != 0;
                double2 *o = reinterpret_cast<double2 *>(out + 4 * k);
                o[0] = make_double2(ar[1], ai[1]);
                o[1] = make_double2(odd_call ? ai[0] : ar[0], odd_call ? ar[NF - 1] : ai[NF - 1]);
                continue;
            }
            double2 *o = reinterpret_cast<double2 *>(out + 2 * (k * P.nfilt));
#pragma unroll
            for (int f = 0; f < NF; ++f)
                if ((NTAPS && STEP) || f < P.nfilt) o[f] = make_double2(ar[f], ai[f]);
        }
    }
}

static int mf_bank_launch(wf_ctx *ctx, const double *d_r_ri, int64_t nsamp, const double *d_taps_ri, int nfilt,
                          int ntaps, int64_t first, int step, int64_t ncols, double *d_out_ri, void *stream,
                          bool noise, double rot_re, double rot_im, double sigma, uint64_t seed,
                          uint64_t stream_id, uint64_t first_index, const uint64_t *d_dyn_index = nullptr,
                          int pack_par0 = -1)
{
    WF_REQUIRE(ctx && d_r_ri && d_taps_ri, "wf_mf_bank_c128: NULL argument");
    WF_REQUIRE(nfilt >= 1 && nfilt <= 8 && ntaps >= 1 && step >= 1 && ncols >= 0 && first >= 0,
               "wf_mf_bank_c128: nfilt %d ntaps %d step %d", nfilt, ntaps, step);
    WF_REQUIRE(nsamp >= ntaps, "wf_mf_bank_c128: input (%lld) shorter than the filter (%d)",
               (long long)nsamp, ntaps);
    WF_REQUIRE(ncols == 0 || first + (ncols - 1) * step < nsamp, "wf_mf_bank_c128: columns run past the input");
    if (ncols == 0) return WF_OK;
    WF_REQUIRE(d_out_ri && (reinterpret_cast<uintptr_t>(d_out_ri) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(d_r_ri) & 15) == 0,
               "wf_mf_bank_c128: device pointers must be 16-byte aligned");
    WF_HIP(hipSetDevice(ctx->device));
    mf_params P;
    P.nsamp = nsamp;
    P.first = first;
    P.ncols = ncols;
    P.step = step;
    P.ntaps = ntaps;
    P.nfilt = nfilt;
    P.c = (ntaps - 1) / 2;
    P.pad = (step % 2 == 0) ? 1 : 0;
    P.rot_re = rot_re;
    P.rot_im = rot_im;
    P.sigma = sigma;
    P.seed = seed;
    P.stream_id = stream_id;
    P.first_index = first_index;
    P.dyn_index = d_dyn_index;
    P.pack_par0 = pack_par0 & 1;
    WF_REQUIRE(pack_par0 < 0 || (noise && step == 8 && nfilt == 3),
               "wf_awgn_mf_bank: packed rows exist for the fused 3-filter, step-8 bank only");
    // slots(ob) = (ob-1)*step + ntaps + pad*(that/step + 1) <= MF_LDS_SLOTS
    int ob = MF_THREADS;
    for (;;) {
        const int span = (ob - 1) * step + ntaps;
        const int slots = span + (P.pad ? span / step + 1 : 0);
        if (slots <= MF_LDS_SLOTS || ob == 1) {
            P.ob = ob;
            P.span = span;
            WF_REQUIRE(span <= MF_MAXI * MF_THREADS, "wf_mf_bank_c128: filter too long for LDS staging (%d taps)", ntaps);
            break;
        }
        ob = ob > 16 ? ob - 16 : ob - 1;
    }
    if (noise && P.ob > 16) {
        // the channel is generated in trips of 2*MF_THREADS samples: a window of 2049 samples (256
        // columns, 9 taps, step 8) would cost a fifth trip for its last pair, and that whole trip
        // lands on one wave.  Give up a few columns when that saves a trip.
        auto trips = [&](int o) { return ((o - 1) * step + ntaps + 1 + 2 * MF_THREADS - 1) / (2 * MF_THREADS); };
        int best = P.ob;
        for (int o = P.ob; o >= P.ob - 16; --o)
            if ((double)o / trips(o) > (double)best / trips(best)) best = o;
        P.ob = best;
        P.span = (best - 1) * step + ntaps;
    }
    // window + the dump slot of the interior staging path; the step-8 channel needs all 256 pad
    // slots (tables), so its dump slot sits behind pad slot 255
    P.dump = P.span + (P.pad ? P.span / step + 1 : 0);
    if (noise && step == 8 && P.dump < 9 * 256) P.dump = 9 * 256;
    P.taps_slot = P.dump + 1;
    const bool long8 = step == 8 && nfilt <= 3 && !(nfilt == 3 && ntaps == 9);   // the <3, *, 8, 0> instantiations
    const int slots = P.dump + 1 + (long8 ? 3 * ntaps : 0);
    P.nblk = (ncols + P.ob - 1) / P.ob;
    const int grid = (int)(P.nblk < 4096 ? P.nblk : 4096);
    hipStream_t s = wf_stream(stream);
    const size_t lds = (size_t)slots * sizeof(double2);
    using kern_t = void (*)(const double *, const double *, double *, mf_params);
    kern_t k;
    if (pack_par0 >= 0) k = ntaps == 9 ? mf_bank_kernel<3, true, 8, 9, true> : mf_bank_kernel<3, true, 8, 0, true>;
    else if (step == 8 && nfilt == 3 && ntaps == 9) k = noise ? mf_bank_kernel<3, true, 8, 9> : mf_bank_kernel<3, false, 8, 9>;
    else if (step == 8 && nfilt <= 3) k = noise ? mf_bank_kernel<3, true, 8, 0> : mf_bank_kernel<3, false, 8, 0>;
    else if (nfilt <= 3) k = noise ? mf_bank_kernel<3, true, 0, 0> : mf_bank_kernel<3, false, 0, 0>;
    else k = noise ? mf_bank_kernel<8, true, 0, 0> : mf_bank_kernel<8, false, 0, 0>;
    if (lds > 48 * 1024)
        WF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(MF_THREADS), lds, s, d_r_ri, d_taps_ri, d_out_ri, P);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

extern "C" int wf_mf_bank_c128(wf_ctx *ctx, const double *d_r_ri, int64_t nsamp, const double *d_taps_ri,
                               int nfilt, int ntaps, int64_t first, int step, int64_t ncols, double *d_out_ri,
                               void *stream)
{
    return mf_bank_launch(ctx, d_r_ri, nsamp, d_taps_ri, nfilt, ntaps, first, step, ncols, d_out_ri, stream, false,
                          1.0, 0.0, 0.0, 0, 0, 0);
}

extern "C" int wf_awgn_mf_bank_c128(wf_ctx *ctx, const double *d_signal_ri, int64_t nsamp, double rot_re,
                                    double rot_im, double sigma, uint64_t seed, uint64_t stream_id,
                                    uint64_t first_index, const double *d_taps_ri, int nfilt, int ntaps,
                                    int64_t first, int step, int64_t ncols, double *d_out_ri, void *stream)
{
    return mf_bank_launch(ctx, d_signal_ri, nsamp, d_taps_ri, nfilt, ntaps, first, step, ncols, d_out_ri, stream, true,
                          rot_re, rot_im, sigma, seed, stream_id, first_index);
}

// Internal: fused channel + bank whose noise counter is first_index + *d_dyn_index (read on
// the device), for the graph-replayed steady state of the streaming link.
// pack_par0 >= 0: detector-packed rows (32 B), pack_par0 = parity of the call index of column 0.
int wf_awgn_mf_bank_dyn(wf_ctx *ctx, const double *d_signal_ri, int64_t nsamp, double rot_re, double rot_im,
                        double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                        const uint64_t *d_dyn_index, const double *d_taps_ri, int nfilt, int ntaps, int64_t first,
                        int step, int64_t ncols, double *d_out_ri, void *stream, int pack_par0)
{
    return mf_bank_launch(ctx, d_signal_ri, nsamp, d_taps_ri, nfilt, ntaps, first, step, ncols, d_out_ri, stream, true,
                          rot_re, rot_im, sigma, seed, stream_id, first_index, d_dyn_index, pack_par0);
}
