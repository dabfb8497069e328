// wf_pipeline.hip — device-resident SOQPSK link: one C-ABI call runs
//   PRBS -> trellis encode -> upsample+FIR -> phase scan + cexp -> derotate + AWGN ->
//   decimating MF bank -> Viterbi -> error count
// entirely out of a caller-provided HBM workspace (the per-waveform body of reference
// examples/soqpsk_detection.py:45-216 without the plotting).  It only sequences the stage
// kernels of this library on one stream; nothing is allocated and nothing comes back to
// the host except the caller's own counters, so the call is graph-capturable and one
// "Monte-Carlo trial block" of the BER sweep / one bench step.
#include "wf_common.h"

// ---------------------------------------------------------------- time axis (a5)
// np.linspace(0, N+1, (N+1)*sps, endpoint=False) (reference waveforms/cpm/modulate.py:81-88)
// evaluates arange(num) * step: one multiply per element, reproduced bit-exactly.
__global__ void time_axis_kernel(int64_t n, double step, double *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        out[k] = (double)k * step;
}

extern "C" int wf_time_axis_f64(wf_ctx *ctx, int64_t n, double step, double *d_out, void *stream)
{
    WF_REQUIRE(ctx && n >= 0, "wf_time_axis_f64: bad argument");
    if (n == 0) return WF_OK;
    WF_REQUIRE(d_out != nullptr, "wf_time_axis_f64: NULL output");
    WF_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(time_axis_kernel, dim3(wf_grid_for(n, 256, 4096)), dim3(256), 0, wf_stream(stream), n,
                       step, d_out);
    WF_LAUNCH_CHECK();
    return WF_OK;
}

// ---------------------------------------------------------------- link
static inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct link_layout {
    int64_t nsym, npts, ncols, first;
    size_t off_bits, off_syms, off_freq, off_sig, off_mf, off_dbits, off_dsyms, total;
};

static link_layout make_layout(int64_t nsym, int sps, int ntaps, int nfilt, int length, int timing_offset)
{
    link_layout L;
    L.nsym = nsym;
    L.npts = wf_fir_out_len(nsym, sps, ntaps);
    // columns n in range(size - length*sps) with (n + timing_offset) % sps == 0
    // (reference examples/soqpsk_detection.py:189-192)
    const int64_t limit = L.npts - (int64_t)length * sps;
    int64_t first = (-(int64_t)timing_offset) % sps;
    if (first < 0) first += sps;
    L.first = first;
    L.ncols = limit > first ? (limit - first + sps - 1) / sps : 0;
    size_t o = 0;
    L.off_bits = o;  o += (size_t)round_up(nsym + 16, 256);
    L.off_syms = o;  o += (size_t)round_up(nsym + 16, 256);
    L.off_freq = o;  o += (size_t)round_up(L.npts * 8, 256);
    L.off_sig = o;   o += (size_t)round_up(L.npts * 16, 256);
    L.off_mf = o;    o += (size_t)round_up(L.ncols * nfilt * 16, 256);
    L.off_dbits = o; o += (size_t)round_up(L.ncols + 16, 256);
    L.off_dsyms = o; o += (size_t)round_up(L.ncols + 16, 256);
    L.total = o;
    return L;
}

// fuse bit 2: the fused channel + bank writes detector-packed rows (4 doubles per call instead of
// 3 complex) and the detector reads those.  Exists for 3-filter banks at 8 samples per symbol.
// Detector chunk warm-up: the caller's, or the detector's own default (32 rows).  A link that
// knows its operating point may ask for less — with a matched bank survivors merge within 12 rows
// from 6 dB up (tools/warmup_scan.py: none of 2.5e6 chunks failed; bench.py asks for 16 rows at its
// 10 dB point) — but a mistimed or mismatched bank at the same Eb/N0 does not (142 chunks of a
// 40 000-symbol burst failed with 16 rows in test_fused_all_other_timing_offsets_and_generic_taps),
// so the library does not guess from sigma.  Every launch proves its output either way.
static int link_warmup(const wf_link_config *cfg) { return cfg->warmup; }

// (the channel + bank kernel's PACK form exists at 8 samples per symbol only: what the streaming link and the
//  fallback of the one-shot link can rely on)
static bool link_packed_rows8(const wf_link_config *cfg)
{
    return (cfg->fuse & 4) && (cfg->fuse & 2) && cfg->sps == 8 && cfg->mf_nfilt == 3;
}

// fuse bit 3 (with bits 0 - 2): modulator + channel + pulse-truncation bank in ONE kernel, detector-packed rows —
// at 8, 10 (the reference's own examples/soqpsk_detection.py:38) and 20 samples per symbol.
static bool link_one_kernel(const wf_link_config *cfg)
{
    if ((cfg->fuse & 15) != 15 || cfg->mf_nfilt != 3 || cfg->sps < 2) return false;
    int64_t first = (-(int64_t)cfg->timing_offset) % cfg->sps;
    if (first < 0) first += cfg->sps;
    return wf_mod_chan_bank_applies(cfg->nsym, 1, cfg->ntaps, cfg->sps, cfg->mf_ntaps, first) != 0;
}

static bool link_packed_rows(const wf_link_config *cfg) { return link_packed_rows8(cfg) || link_one_kernel(cfg); }

// fuse bit 5 (with the one-kernel front end): the detector and the error count of a block run on the context's side
// stream, so that they overlap the front end of the NEXT wf_link_run on the same context — consecutive blocks are
// independent trial blocks, and the front-end kernel (vector-issue-bound) leaves what the detector (HBM-bound, one wave
// per SIMD) needs.  The workspace then holds two sets of intermediates, used alternately.
static bool link_pipelined(const wf_link_config *cfg) { return (cfg->fuse & 32) && link_one_kernel(cfg); }

// d_mf_factor promises d_mf_taps[s] == sum_k G[s][k] b_k (two real filters b_k, a 3 x 2 complex combination G): the
// one-kernel front end then runs the two real filters.  Checked on a host copy to 1e-12 of the largest tap (what the
// Python link checks before it hands a factorisation over; wf_promise_verified).  Only the long-bank form reads the factors.
static bool link_factor_matches(const unsigned char *const *host, const size_t *, const void *arg)
{
    const int nt = *static_cast<const int *>(arg);
    const double *taps = reinterpret_cast<const double *>(host[0]);          // [3][nt] (re, im)
    const double *b = reinterpret_cast<const double *>(host[1]);             // b_0[nt], b_1[nt], G[3][2] (re, im)
    const double *G = b + 2 * nt;
    double big = 0.0, worst = 0.0;
    for (int s = 0; s < 3; ++s)
        for (int t = 0; t < nt; ++t) {
            const double re = G[4 * s] * b[t] + G[4 * s + 2] * b[nt + t], im = G[4 * s + 1] * b[t] + G[4 * s + 3] * b[nt + t];
            const double tr = taps[2 * (s * nt + t)], ti = taps[2 * (s * nt + t) + 1];
            const double d = hypot(re - tr, im - ti), m = hypot(tr, ti);
            if (!(d == d)) return false;
            if (d > worst) worst = d;
            if (m > big) big = m;
        }
    return worst <= 1e-12 * big;
}
static int link_check_factor(wf_ctx *ctx, const wf_link_config *cfg, void *stream)
{
    if (!cfg->d_mf_factor || cfg->mf_ntaps == cfg->sps + 1 || cfg->mf_nfilt != 3 || cfg->mf_ntaps < 1) return WF_OK;
    const int nt = cfg->mf_ntaps;
    const void *ptrs[2] = {cfg->d_mf_taps, cfg->d_mf_factor};
    const size_t nb[2] = {(size_t)3 * nt * 16, (size_t)(2 * nt + 12) * 8};
    return wf_promise_verified(ctx, 2, ptrs, nb, 2, stream, link_factor_matches, &nt,
                               "wf_link_config.d_mf_factor does not reproduce d_mf_taps (taps[s] != sum_k G[s][k] b_k to 1e-12)");
}

extern "C" int64_t wf_link_workspace_bytes(const wf_link_config *cfg)
{
    if (!cfg || cfg->nsym < 1 || cfg->sps < 1) return -1;
    const int64_t one = (int64_t)make_layout(cfg->nsym, cfg->sps, cfg->ntaps, cfg->mf_nfilt, 2, cfg->timing_offset).total;
    return link_pipelined(cfg) ? 2 * round_up(one, 256) : one;       // fuse bit 5: two sets of intermediates
}

// The stream a pipelined link's front-end kernel runs on: every compute unit but the first `reserve` (hipExtStreamCreateWithCUMask),
// so that the small kernels of the NEXT block's prologue — which fit neither the registers (4 x 128 of a SIMD's 512) nor the LDS
// (4 x 40 KB of a CU's 160) the front end leaves — and the previous block's detector always find a few CUs of their own.
// (Such a stream has the default flags: it orders itself against the legacy NULL stream like any stream a caller creates.)
static hipStream_t link_masked_stream(const wf_ctx *ctx, int reserve)
{
    hipStream_t s = nullptr;
    if (reserve > 0 && reserve < ctx->cus) {
        const int words = (ctx->cus + 31) / 32;
        std::vector<uint32_t> mask((size_t)words, 0u);
        for (int cu = reserve; cu < ctx->cus; ++cu) mask[(size_t)(cu / 32)] |= 1u << (cu % 32);
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()) == hipSuccess) return s;
        (void)hipGetLastError();
        s = nullptr;
    }
    if (hipStreamCreateWithFlags(&s, hipStreamDefault) != hipSuccess) return nullptr;
    return s;
}

extern "C" int wf_link_run(wf_ctx *ctx, const wf_link_config *cfg, void *d_workspace, int64_t workspace_bytes,
                           int64_t *d_counts, int64_t *h_compared, void *stream)
{
    WF_REQUIRE(ctx && cfg && d_workspace && d_counts, "wf_link_run: NULL argument");
    WF_REQUIRE(cfg->nsym >= 1 && cfg->sps >= 1 && cfg->mf_nfilt == 3, "wf_link_run: bad configuration");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_workspace) & 255) == 0, "wf_link_run: workspace must be 256-byte aligned");
    const int length = 2;
    const link_layout L = make_layout(cfg->nsym, cfg->sps, cfg->ntaps, cfg->mf_nfilt, length, cfg->timing_offset);
    WF_REQUIRE((int64_t)L.total <= workspace_bytes, "wf_link_run: workspace too small (%lld < %lld)",
               (long long)workspace_bytes, (long long)L.total);
    WF_REQUIRE(L.npts >= cfg->mf_ntaps, "wf_link_run: burst shorter than the matched filter");
    char *w = static_cast<char *>(d_workspace);
    const bool piped = link_pipelined(cfg) && L.ncols > 0;
    void *s_pro = stream, *s_main = stream;       // where the prologue / the front-end kernel go (a pipelined link: library streams)
    bool ahead = false;
    int slot = -1;
    if (piped) {
        const int64_t set_bytes = round_up((int64_t)L.total, 256);
        WF_REQUIRE(2 * set_bytes <= workspace_bytes, "wf_link_run: fuse bit 5 needs two sets of intermediates (%lld bytes)", (long long)(2 * set_bytes));
        WF_HIP(hipSetDevice(ctx->device));
        if (!ctx->pipe_stream) {
            hipStream_t ps;
            WF_HIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
            ctx->pipe_stream = ps;
            WF_HIP(hipEventCreateWithFlags(&ctx->pipe_front, hipEventDisableTiming));
            for (int k = 0; k < 2; ++k) WF_HIP(hipEventCreateWithFlags(&ctx->pipe_done[k], hipEventDisableTiming));
        }
        w += (int64_t)ctx->pipe_set * set_bytes;
        // Round 6: three library streams per pipelined link.  A block's PROLOGUE (PRBS + precoder, tile sums, tile scan: ~60 us of
        // small kernels that used to sit between two front ends on the caller's stream) goes to pipe_pro and depends only on its
        // set of intermediates being free — the host queues block k + 1 while block k's front end runs, so it runs BESIDE that front
        // end; the front-end kernel goes to pipe_main (CU-masked: see link_masked_stream) behind its own prologue and the previous
        // front end; detector + count to pipe_stream as before.  The carries live in one of two scratch sets (slot = the set).
        // MEASURED AND NOT KEPT as the default (option WF_OPT_PIPE_RESERVE_CUS, 0 = off): the front ends then follow each other
        // with 12 us between them, but each runs 404 - 475 us instead of 380 — what the detector and the prologue kernels take from
        // it when they run beside it is what they cost between two front ends (the chip's work is conserved: the front end fills
        // every SIMD's registers and every CU's LDS, nothing co-resides with it) — 0.457 ms per block as shipped, 0.474 without a
        // mask, 0.50 with 8 CUs masked (profiles/r06_ab_prologue_ahead_cu_mask.log, r06_timeline_soqpsk_prologue_ahead.txt).
        const int64_t rsv = ctx->opt[WF_OPT_PIPE_RESERVE_CUS];
        ahead = rsv != 0;
        if (ahead) {
            if (!ctx->pipe_pro) {
                hipStream_t sp;
                WF_HIP(hipStreamCreateWithFlags(&sp, hipStreamDefault));       // (default flags: ordered behind what the caller queued on the NULL stream)
                ctx->pipe_pro = sp;
                ctx->pipe_main = link_masked_stream(ctx, rsv < 0 ? 0 : (int)rsv);
                WF_REQUIRE(ctx->pipe_main != nullptr, "wf_link_run: could not create the front end's stream");
                WF_HIP(hipEventCreateWithFlags(&ctx->pipe_call, hipEventDisableTiming));
                WF_HIP(hipEventCreateWithFlags(&ctx->pipe_pro_done, hipEventDisableTiming));
            }
            s_pro = ctx->pipe_pro;
            s_main = ctx->pipe_main;
            slot = ctx->pipe_set;
            if (stream) {          // a caller's own stream: what it queued there comes first (the NULL stream orders itself, and recording on it would wait for the front end in flight)
                WF_HIP(hipEventRecord(ctx->pipe_call, wf_stream(stream)));
                WF_HIP(hipStreamWaitEvent(wf_stream(s_pro), ctx->pipe_call, 0));
            }
            if (ctx->pipe_done_valid[ctx->pipe_set]) WF_HIP(hipStreamWaitEvent(wf_stream(s_pro), ctx->pipe_done[ctx->pipe_set], 0));
        } else if (ctx->pipe_done_valid[ctx->pipe_set]) {
            // the set was last used two blocks ago: that block's detector and counter must be done with it
            WF_HIP(hipStreamWaitEvent(wf_stream(stream), ctx->pipe_done[ctx->pipe_set], 0));
        }
    } else {
        const int rj = wf_link_join_internal(ctx, stream);      // (a context that ran pipelined blocks before: ordinary stream order from here on)
        if (rj) return rj;
    }
    uint8_t *bits = reinterpret_cast<uint8_t *>(w + L.off_bits);
    int8_t *syms = reinterpret_cast<int8_t *>(w + L.off_syms);
    double *freq = reinterpret_cast<double *>(w + L.off_freq);
    double *sig = reinterpret_cast<double *>(w + L.off_sig);
    double *mf = reinterpret_cast<double *>(w + L.off_mf);
    uint8_t *dbits = reinterpret_cast<uint8_t *>(w + L.off_dbits);
    int8_t *dsyms = reinterpret_cast<int8_t *>(w + L.off_dsyms);

    // SOQPSKTrellis4x2 / 4x2DiffEncoded as dense [column][state][input] tables
    // (reference waveforms/cpm/trellis/model.py:205-258).
    uint8_t next[2][4][2];
    int8_t outp[2][4][2];
    static const int8_t kOut[2][8] = {{0, 2, 0, -2, -2, 0, 2, 0}, {0, -2, 2, 0, 0, 2, -2, 0}};
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 8; ++b) {
            const int s = b >> 1;
            const int e = c == 0 ? (s & 1) + 2 * (b & 1) : (s & 2) + (b & 1);
            const int flip = cfg->differential ? (c == 0 ? (s >> 1) : (s & 1)) : 0;
            const int inp = (b & 1) ^ flip;
            next[c][s][inp] = (uint8_t)e;
            outp[c][s][inp] = kOut[c][b];
        }
    hipEvent_t *ev = nullptr;
    if (cfg->event_slot >= 0) {
        WF_REQUIRE(cfg->event_slot < WF_LINK_EVENT_SLOTS, "wf_link_run: event_slot %d", cfg->event_slot);
        if (!ctx->events) {
            ctx->events = new hipEvent_t[WF_LINK_EVENT_SLOTS * (WF_LINK_STAGES + 1)];
            for (int k = 0; k < WF_LINK_EVENT_SLOTS * (WF_LINK_STAGES + 1); ++k) WF_HIP(hipEventCreate(&ctx->events[k]));
        }
        ev = ctx->events + cfg->event_slot * (WF_LINK_STAGES + 1);
    }
#define MARKS(k, s) do { if (ev) WF_HIP(hipEventRecord(ev[k], wf_stream(s))); } while (0)
#define MARK(k) MARKS(k, stream)
    int rc;
    MARKS(0, s_pro);
    // PRBS + precoder: two launches when the burst fits the scan-free form (wf_soqpsk_prbs_encode), else the generic four
    rc = (cfg->fuse & 16) ? 1 : wf_soqpsk_prbs_encode(ctx, cfg->degree, cfg->mask, cfg->state, cfg->skip, &next[0][0][0], &outp[0][0][0], bits,
                                                       cfg->nsym, syms, s_pro, ev ? (void *)ev[1] : nullptr);
    if (rc < 0) return rc;
    if (rc == 1) {
        if ((rc = wf_lfsr_generate(ctx, cfg->degree, cfg->mask, cfg->state, cfg->skip, bits, cfg->nsym, nullptr, s_pro))) return rc;
        MARKS(1, s_pro);
        if ((rc = wf_fsm_encode(ctx, &next[0][0][0], &outp[0][0][0], 2, 4, 1, bits, cfg->nsym, 0, 0, syms, nullptr, s_pro))) return rc;
    }
    if (!ahead) MARKS(2, s_pro);
    // fuse bit 3 (with bits 1 and 2 in effect): modulator, channel and bank in ONE kernel — the clean
    // baseband samples never exist in HBM.  Outside that kernel's envelope the bits below apply.
    bool fused_all = false;
    if (link_one_kernel(cfg) && L.ncols > 0) {
        wf_mcb_opts mo;
        mo.pam_factor = cfg->d_mf_factor;             // (a long bank handed over in factored form: two real filters + a 3 x 2 combination — checked)
        mo.scratch_slot = slot;
        if ((rc = link_check_factor(ctx, cfg, stream))) return rc;
        for (int stage = ahead ? 1 : 3; stage <= (ahead ? 2 : 3); ++stage) {
            // (a pipelined link: the carry kernels with the prologue, the main kernel on the front end's stream behind them)
            if (ahead && stage == 2) {
                // stage events of such a block: "encode" = precoder + the two carry kernels (the prologue's tail), "fir" = the wait
                // between the prologue's end and the main kernel's start (no kernel), "phase" = the main kernel
                MARKS(2, s_pro);
                WF_HIP(hipEventRecord(ctx->pipe_pro_done, wf_stream(s_pro)));
                WF_HIP(hipStreamWaitEvent(wf_stream(s_main), ctx->pipe_pro_done, 0));
                MARKS(3, s_main);
            }
            rc = wf_mod_chan_bank_packed(ctx, syms, cfg->nsym, cfg->d_h, 1, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, cfg->d_mf_taps,
                                         cos(-M_PI / 4), sin(-M_PI / 4), cfg->sigma, cfg->seed, cfg->stream_id, 0, L.first, L.ncols, 0,
                                         mf, stage == 1 ? s_pro : s_main, cfg->mf_ntaps, &mo, stage);
            if (rc < 0) return rc;
            WF_REQUIRE(rc == 0, "wf_link_run: internal: the one-kernel front end refused a configuration wf_mod_chan_bank_applies accepted");
        }
        fused_all = true;
    }
    if (fused_all) {
        if (!ahead) MARKS(3, s_main);                                    // the "fir" slot times the whole fused kernel (the "phase" slot when the prologue runs ahead: above)
        MARKS(4, s_main); MARKS(5, s_main);
        void *back = stream;                  // the stream the detector and the counter run on
        if (piped) {
            back = ctx->pipe_stream;
            WF_HIP(hipEventRecord(ctx->pipe_front, wf_stream(s_main)));
            WF_HIP(hipStreamWaitEvent(wf_stream(back), ctx->pipe_front, 0));
        }
#define MARKB(k) do { if (ev) WF_HIP(hipEventRecord(ev[k], wf_stream(back))); } while (0)
        MARKB(6);
        rc = wf_viterbi4_detect_packed(ctx, mf, L.ncols, cfg->differential, link_warmup(cfg), dbits, dsyms, nullptr, back);
        if (rc) return rc;
        MARKB(7);
        int64_t m_ = L.ncols - length;
        if (m_ > cfg->nsym) m_ = cfg->nsym;
        if (m_ < 0) m_ = 0;
        if ((rc = wf_count_errors(ctx, dsyms + length, syms, dbits + length, bits, m_, d_counts, back))) return rc;
        MARKB(8);
#undef MARKB
        if (piped) {
            WF_HIP(hipEventRecord(ctx->pipe_done[ctx->pipe_set], wf_stream(back)));
            ctx->pipe_done_valid[ctx->pipe_set] = true;
            ctx->pipe_set ^= 1;
        }
        if (h_compared) *h_compared = m_;
        return WF_OK;
    }
    bool fused_mod = false;
    if (cfg->fuse & 1) {
        rc = wf_cpm_modulate_c128(ctx, syms, cfg->nsym, cfg->d_h, 1, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, sig, stream);
        if (rc < 0) return rc;
        fused_mod = rc == 0;
    }
    if (fused_mod) {
        MARK(3);   // the "fir" slot times the fused modulator, the "phase" slot is empty
    } else {
        if ((rc = wf_upsample_fir_f64(ctx, syms, cfg->nsym, cfg->d_h, 1, cfg->d_pulse, cfg->ntaps, cfg->sps, freq, stream))) return rc;
        MARK(3);
        if ((rc = wf_phase_cexp_f64(ctx, freq, L.npts, cfg->sps, M_PI / 4, 0.0, sig, nullptr, stream))) return rc;
    }
    MARK(4);
    const double rot_re = cos(-M_PI / 4), rot_im = sin(-M_PI / 4);
    const bool fused_chan = (cfg->fuse & 2) && L.ncols > 0;
    // modulated *= exp(-j pi/4); received = modulated + noise   (in place, or inside the bank)
    if (!fused_chan)
        if ((rc = wf_awgn_c128(ctx, sig, L.npts, rot_re, rot_im, cfg->sigma, cfg->seed, cfg->stream_id, 0, sig, stream))) return rc;
    MARK(5);
    // drop the first `length` detector outputs, compare min_size elements
    // (reference examples/soqpsk_detection.py:201-209)
    int64_t m = L.ncols - length;
    if (m > cfg->nsym) m = cfg->nsym;
    if (m < 0) m = 0;
    const bool fused_count = false;
    const bool packed = fused_chan && link_packed_rows8(cfg);
    if (L.ncols > 0) {
        if (packed)   // detector-packed rows: 4 doubles per call (call index of column 0 is 0: even)
            rc = wf_awgn_mf_bank_dyn(ctx, sig, L.npts, rot_re, rot_im, cfg->sigma, cfg->seed, cfg->stream_id, 0, nullptr,
                                     cfg->d_mf_taps, cfg->mf_nfilt, cfg->mf_ntaps, L.first, cfg->sps, L.ncols, mf, stream, 0);
        else if (fused_chan)
            rc = wf_awgn_mf_bank_c128(ctx, sig, L.npts, rot_re, rot_im, cfg->sigma, cfg->seed, cfg->stream_id, 0,
                                      cfg->d_mf_taps, cfg->mf_nfilt, cfg->mf_ntaps, L.first, cfg->sps, L.ncols, mf, stream);
        else
            rc = wf_mf_bank_c128(ctx, sig, L.npts, cfg->d_mf_taps, cfg->mf_nfilt, cfg->mf_ntaps, L.first, cfg->sps,
                                 L.ncols, mf, stream);
        if (rc) return rc;
        MARK(6);
        if (fused_count)
            rc = wf_viterbi4_detect_count(ctx, mf, L.ncols, cfg->differential, link_warmup(cfg), dbits, dsyms, bits, syms,
                                          length, m, d_counts, stream);
        else if (packed)
            rc = wf_viterbi4_detect_packed(ctx, mf, L.ncols, cfg->differential, link_warmup(cfg), dbits, dsyms, nullptr, stream);
        else
            rc = wf_viterbi4_detect(ctx, mf, L.ncols, cfg->differential, link_warmup(cfg), dbits, dsyms, nullptr, stream);
        if (rc) return rc;
    } else {
        MARK(6);
    }
    MARK(7);
    if (m > 0 && !fused_count)
        if ((rc = wf_count_errors(ctx, dsyms + length, syms, dbits + length, bits, m, d_counts, stream))) return rc;
    MARK(8);
#undef MARK
    if (h_compared) *h_compared = m;
    return WF_OK;
}

extern "C" int wf_link_stage_ms(wf_ctx *ctx, int event_slot, float *h_ms)
{
    WF_REQUIRE(ctx && h_ms && event_slot >= 0 && event_slot < WF_LINK_EVENT_SLOTS && ctx->events,
               "wf_link_stage_ms: no events recorded in slot %d", event_slot);
    hipEvent_t *ev = ctx->events + event_slot * (WF_LINK_STAGES + 1);
    WF_HIP(hipEventSynchronize(ev[WF_LINK_STAGES]));
    for (int k = 0; k < WF_LINK_STAGES; ++k) WF_HIP(hipEventElapsedTime(&h_ms[k], ev[k], ev[k + 1]));
    return WF_OK;
}

extern "C" int wf_link_layout(const wf_link_config *cfg, int64_t *info8)
{
    if (!cfg || !info8 || cfg->nsym < 1 || cfg->sps < 1) return WF_ERR_VALUE;
    const link_layout L = make_layout(cfg->nsym, cfg->sps, cfg->ntaps, cfg->mf_nfilt, 2, cfg->timing_offset);
    info8[0] = L.ncols; info8[2] = (int64_t)L.off_dbits; info8[3] = (int64_t)L.off_dsyms;
    info8[4] = (int64_t)L.off_sig; info8[6] = L.npts; info8[7] = (int64_t)L.off_mf;
    info8[1] = link_one_kernel(cfg) ? 1 : 0;                          // modulator + channel + bank run as ONE kernel
    info8[5] = link_packed_rows(cfg) ? 32 : 16 * cfg->mf_nfilt;       // bytes per matched-filter row at off(mf)
    return WF_OK;
}

// ------------------------------------------------------------------------------------------
// Streaming link (BASELINE config 5): the same chain over a stream of cfg->nsym symbols, one
// chunk of `chunk_symbols` detector calls per call.  Chunk c detects calls [c*B, (c+1)*B);
// what it needs from its neighbours comes from (a) re-generating a halo — PRBS by
// leap-ahead, noise by absolute counter, one modulator tile of samples and sym_per_tile + 48
// symbols on either side — and (b) a 512-byte device-resident carry block: Viterbi state
// (32 doubles), encoder state at the next window start, modulator phase carry (62-bit
// fixed point) of the next window's first tile.  Decisions and counts equal the one-shot
// wf_link_run over the whole stream.
struct stream_layout {
    int64_t N, B, c, tile_len, spt, ntiles_total, halo, npts, ncols_total, first, m_total;
    int64_t k_lo, ncols, ws, nloc, tile_lo, ntiles, out_origin, local_len, ws_next, q_out_tile;
    size_t off_bits, off_syms, off_sig, off_mf, off_dbits, off_dsyms, total;
    bool ok;
};

static stream_layout make_stream_layout(const wf_link_config *cfg, int64_t B, int64_t c)
{
    stream_layout S;
    const int length = 2;
    S.N = cfg->nsym; S.B = B; S.c = c;
    if (cfg->sps < 2 || cfg->sps > 256 || cfg->ntaps < 1 || cfg->nsym < 1 || cfg->mf_ntaps < 1) {
        S = stream_layout{};   // everything zero, ok = false: nothing below may divide by sps / spt
        return S;
    }
    S.ok = wf_mod_tile_geometry(cfg->sps, cfg->ntaps, cfg->nsym, &S.tile_len, &S.spt, &S.ntiles_total) == 0;
    S.halo = round_up(S.spt + 48, 16);
    S.npts = wf_fir_out_len(S.N, cfg->sps, cfg->ntaps);
    const int64_t limit = S.npts - (int64_t)length * cfg->sps;
    int64_t first = (-(int64_t)cfg->timing_offset) % cfg->sps;
    if (first < 0) first += cfg->sps;
    S.first = first;
    S.ncols_total = limit > first ? (limit - first + cfg->sps - 1) / cfg->sps : 0;
    S.m_total = S.ncols_total - length < S.N ? S.ncols_total - length : S.N;
    if (S.m_total < 0) S.m_total = 0;
    S.ok = S.ok && B > 0 && B % S.spt == 0 && B % 128 == 0 && B >= 4 * S.halo &&
           S.tile_len >= cfg->mf_ntaps + 2 * cfg->sps;
    S.k_lo = c * B;
    const int64_t k_hi = S.k_lo + B < S.ncols_total ? S.k_lo + B : S.ncols_total;
    S.ncols = k_hi > S.k_lo ? k_hi - S.k_lo : 0;
    S.ws = c * B - S.halo > 0 ? c * B - S.halo : 0;
    const int64_t we = (c + 1) * B + S.halo < S.N ? (c + 1) * B + S.halo : S.N;
    S.nloc = we > S.ws ? we - S.ws : 0;
    S.tile_lo = S.spt > 0 && c * B / S.spt - 1 > 0 ? c * B / S.spt - 1 : 0;
    int64_t tile_hi = S.spt > 0 ? (c + 1) * B / S.spt + 1 : 0;
    if (tile_hi > S.ntiles_total) tile_hi = S.ntiles_total;
    S.ntiles = tile_hi > S.tile_lo ? tile_hi - S.tile_lo : 0;
    S.out_origin = S.tile_lo * S.tile_len;
    const int64_t hi = tile_hi * S.tile_len < S.npts ? tile_hi * S.tile_len : S.npts;
    S.local_len = hi > S.out_origin ? hi - S.out_origin : 0;
    S.ws_next = (c + 1) * B - S.halo > 0 ? (c + 1) * B - S.halo : 0;
    const int64_t next_tile_lo = S.spt > 0 ? (c + 1) * B / S.spt - 1 : 0;
    S.q_out_tile = (next_tile_lo >= S.tile_lo && next_tile_lo < tile_hi) ? next_tile_lo - S.tile_lo : -1;
    size_t o = 0;
    const int64_t win = B + 2 * S.halo + 32;
    S.off_bits = o;  o += (size_t)round_up(win, 256);
    S.off_syms = o;  o += (size_t)round_up(win, 256);
    S.off_sig = o;   o += (size_t)round_up((B * cfg->sps + 2 * S.tile_len) * 16, 256);
    S.off_mf = o;    o += (size_t)round_up(B * cfg->mf_nfilt * 16, 256);
    S.off_dbits = o; o += (size_t)round_up(B + 16, 256);
    S.off_dsyms = o; o += (size_t)round_up(B + 16, 256);
    S.total = o;
    return S;
}

extern "C" int64_t wf_link_stream_workspace_bytes(const wf_link_config *cfg, int64_t chunk_symbols)
{
    if (!cfg || cfg->nsym < 1 || cfg->sps < 2) return -1;
    const stream_layout S = make_stream_layout(cfg, chunk_symbols, 0);
    return S.ok ? (int64_t)S.total : -1;
}

extern "C" int wf_link_stream_layout(const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index, int64_t *info8)
{
    if (!cfg || !info8 || cfg->nsym < 1 || cfg->sps < 2) return WF_ERR_VALUE;
    const stream_layout S = make_stream_layout(cfg, chunk_symbols, chunk_index);
    if (!S.ok) return WF_ERR_VALUE;
    info8[0] = S.ncols; info8[1] = S.k_lo; info8[2] = (int64_t)S.off_dbits; info8[3] = (int64_t)S.off_dsyms;
    info8[4] = (int64_t)S.off_sig; info8[5] = S.out_origin; info8[6] = S.local_len; info8[7] = (int64_t)S.off_mf;
    return WF_OK;
}

__global__ void stream_advance_kernel(uint64_t *word, uint64_t by)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *word += by;
}

static int stream_chunk_impl(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                             void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                             int64_t *h_compared, void *stream, bool steady, int phases = 7);
extern "C" int wf_link_stream_interior(const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index);

extern "C" int wf_link_stream_chunk(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                                    void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                                    int64_t *h_compared, void *stream)
{
    return stream_chunk_impl(ctx, cfg, chunk_symbols, chunk_index, d_state, d_workspace, workspace_bytes, d_counts,
                             h_compared, stream, false);
}

// 1 if chunk `chunk_index` issues exactly the launches of chunk 1 (full chunk, nothing clipped
// by the ends of the stream, full comparison range), i.e. can be served by a replay of the
// steady-state graph; 0 otherwise.
extern "C" int wf_link_stream_interior(const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index)
{
    if (!cfg || cfg->nsym < 1 || cfg->sps < 2 || chunk_index < 1) return 0;
    const stream_layout A = make_stream_layout(cfg, chunk_symbols, 1);
    const stream_layout S = make_stream_layout(cfg, chunk_symbols, chunk_index);
    if (!A.ok || !S.ok) return 0;
    const int64_t sym_last = S.k_lo + S.ncols - 2;   // one past the last reference symbol compared
    return A.ncols == chunk_symbols && S.ncols == chunk_symbols && S.nloc == A.nloc && S.ntiles == A.ntiles &&
           S.local_len == A.local_len && A.tile_lo > 0 && S.ws - S.k_lo == A.ws - A.k_lo &&
           S.out_origin - S.k_lo * cfg->sps == A.out_origin - A.k_lo * cfg->sps && sym_last <= S.m_total &&
           S.q_out_tile == A.q_out_tile && (S.ws_next - S.ws) == (A.ws_next - A.ws);
}

extern "C" int wf_link_stream_steady(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, void *d_state,
                                     void *d_workspace, int64_t workspace_bytes, int64_t *d_counts, int64_t *h_compared,
                                     void *stream)
{
    return stream_chunk_impl(ctx, cfg, chunk_symbols, 1, d_state, d_workspace, workspace_bytes, d_counts, h_compared,
                             stream, true);
}

// The steady-state launch sequence in parts (see wf_link_stream_chunk_phase), for capturing a PIPELINE of
// interior chunks on two streams as one hipGraph: every part advances its own position word at its end.
extern "C" int wf_link_stream_steady_phase(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, void *d_state,
                                           void *d_workspace, int64_t workspace_bytes, int64_t *d_counts, int64_t *h_compared,
                                           int phases, void *stream)
{
    WF_REQUIRE(phases >= 1 && phases <= 7, "wf_link_stream_steady_phase: phases %d", phases);
    return stream_chunk_impl(ctx, cfg, chunk_symbols, 1, d_state, d_workspace, workspace_bytes, d_counts, h_compared,
                             stream, true, phases);
}

// Parts of a chunk for callers that pipeline chunks on two streams (own workspace AND own wf_ctx per
// stream): bit 0 = PRBS, encoder, modulator carries (needs the encoder / phase carries the previous
// chunk's bit-0 part left in d_state); bit 2 = modulator + channel + bank (needs only this chunk's
// bit-0 part; with the separate kernels — fuse without bit 3 — also the previous chunk's bit-2 part);
// bit 1 = detector + error count (needs the detector carry of the previous chunk's bit-1 part).
// phases = 7: the whole chunk (wf_link_stream_chunk).
extern "C" int wf_link_stream_chunk_phase(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                                          void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                                          int64_t *h_compared, int phases, void *stream)
{
    WF_REQUIRE(phases >= 1 && phases <= 7, "wf_link_stream_chunk_phase: phases %d", phases);
    return stream_chunk_impl(ctx, cfg, chunk_symbols, chunk_index, d_state, d_workspace, workspace_bytes, d_counts, h_compared,
                             stream, false, phases);
}

static int stream_chunk_impl(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                             void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                             int64_t *h_compared, void *stream, bool steady, int phases)
{
    WF_REQUIRE(ctx && cfg && d_state && d_workspace && d_counts, "wf_link_stream_chunk: NULL argument");
    WF_REQUIRE(cfg->nsym >= 1 && cfg->sps >= 2 && cfg->mf_nfilt == 3 && chunk_index >= 0, "wf_link_stream_chunk: bad configuration");
    WF_REQUIRE((reinterpret_cast<uintptr_t>(d_workspace) & 255) == 0 && (reinterpret_cast<uintptr_t>(d_state) & 15) == 0,
               "wf_link_stream_chunk: workspace must be 256-byte aligned");
    const int length = 2;
    const stream_layout S = make_stream_layout(cfg, chunk_symbols, chunk_index);
    WF_REQUIRE(S.ok, "wf_link_stream_chunk: chunk of %lld symbols is not a multiple of the modulator tile (%lld symbols) "
               "and of 128, is shorter than 4 halos, or the pulse is outside the fused modulator",
               (long long)chunk_symbols, (long long)S.spt);
    WF_REQUIRE((int64_t)S.total <= workspace_bytes, "wf_link_stream_chunk: workspace too small");
    if (h_compared) *h_compared = 0;
    if (S.ncols == 0) return WF_OK;
    char *w = static_cast<char *>(d_workspace);
    uint8_t *bits = reinterpret_cast<uint8_t *>(w + S.off_bits);
    int8_t *syms = reinterpret_cast<int8_t *>(w + S.off_syms);
    double *sig = reinterpret_cast<double *>(w + S.off_sig);
    double *mf = reinterpret_cast<double *>(w + S.off_mf);
    uint8_t *dbits = reinterpret_cast<uint8_t *>(w + S.off_dbits);
    int8_t *dsyms = reinterpret_cast<int8_t *>(w + S.off_dsyms);
    char *carry = static_cast<char *>(d_state);
    double *vit_state = reinterpret_cast<double *>(carry);
    int *enc_state = reinterpret_cast<int *>(carry + 256);
    uint64_t *q_phase = reinterpret_cast<uint64_t *>(carry + 264);
    // steady-state form: the launch sequence of chunk 1, with the two quantities that differ
    // between interior chunks — PRBS position and noise counter — read from the carry block
    // (words at +272 / +280) and advanced by one chunk at the end.  Identical launches every
    // time => capturable once as a hipGraph and replayed for every interior chunk.
    uint64_t *dyn = reinterpret_cast<uint64_t *>(carry + 272);
    if (steady)
        WF_REQUIRE(wf_link_stream_interior(cfg, chunk_symbols, 1), "wf_link_stream_steady: the stream has no interior chunk of this size");

    uint8_t next[2][4][2];
    int8_t outp[2][4][2];
    static const int8_t kOut[2][8] = {{0, 2, 0, -2, -2, 0, 2, 0}, {0, -2, 2, 0, 0, 2, -2, 0}};
    for (int c = 0; c < 2; ++c)
        for (int b = 0; b < 8; ++b) {
            const int s = b >> 1;
            const int e = c == 0 ? (s & 1) + 2 * (b & 1) : (s & 2) + (b & 1);
            const int flip = cfg->differential ? (c == 0 ? (s >> 1) : (s & 1)) : 0;
            next[c][s][(b & 1) ^ flip] = (uint8_t)e;
            outp[c][s][(b & 1) ^ flip] = kOut[c][b];
        }
    int rc = WF_OK;
    const bool packed = link_packed_rows8(cfg);     // (the stream's one-kernel windows are an sps-8 path)
    const double rot_re = cos(-M_PI / 4), rot_im = sin(-M_PI / 4);
    WF_REQUIRE(!steady || (cfg->fuse & 2), "wf_link_stream_steady needs the fused channel (fuse bit 1)");
    if (phases & 1) {
        if ((rc = wf_lfsr_generate_dyn(ctx, cfg->degree, cfg->mask, cfg->state, cfg->skip + (uint64_t)S.ws, steady ? dyn : nullptr,
                                       bits, S.nloc, nullptr, stream))) return rc;
        const int64_t at = S.ws_next - S.ws;
        if ((rc = wf_fsm_encode_core(ctx, &next[0][0][0], &outp[0][0][0], 2, 4, 1, bits, S.nloc, S.ws, 0, enc_state, syms, nullptr,
                                     at <= S.nloc ? enc_state : nullptr, at, stream))) return rc;
        if (steady) {   // the PRBS position of the next chunk's part 1 (each part advances ITS word: parts of different chunks overlap)
            hipLaunchKernelGGL(stream_advance_kernel, dim3(1), dim3(64), 0, wf_stream(stream), dyn, (uint64_t)chunk_symbols);
            WF_LAUNCH_CHECK();
        }
    }
    // fuse bit 3: modulator + channel + bank of the chunk's tiles in one kernel (no samples in HBM);
    // the tile before the chunk is processed too — its last column is the chunk's first.  Its two
    // carry kernels (stage 1) belong to part bit 0, the main kernel (stage 2) to part bit 2.
    bool fused_all = false;
    // (any bank the kernel takes at 8 samples per symbol: the 9-tap pulse-truncation bank, or an odd-length bank of up to 73 taps)
    if ((cfg->fuse & 8) && packed && (phases & 5) && wf_mod_chan_bank_applies(S.N, 1, cfg->ntaps, cfg->sps, cfg->mf_ntaps, S.first)) {
        const int stage = ((phases & 1) ? 1 : 0) | ((phases & 4) ? 2 : 0);
        wf_mcb_opts mo;
        mo.pam_factor = cfg->d_mf_factor;
        if ((rc = link_check_factor(ctx, cfg, stream))) return rc;
        rc = wf_mod_chan_bank_window(ctx, syms, S.ws, S.nloc, S.N, cfg->d_h, 1, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4, S.tile_lo,
                                     S.ntiles, q_phase, q_phase, S.q_out_tile, cfg->d_mf_taps, rot_re, rot_im, cfg->sigma, cfg->seed,
                                     cfg->stream_id, 0, steady ? dyn + 1 : nullptr, S.first, S.k_lo, S.ncols, 0, mf, stream, 0, 1, stage,
                                     cfg->mf_ntaps, &mo);
        if (rc < 0) return rc;
        fused_all = rc == 0;
    }
    if ((phases & 4) && !fused_all) {
        // separate kernels: the modulator call computes its own carries (so this part then depends
        // on the previous chunk's bit-2 part through the phase carry)
        if ((rc = wf_cpm_modulate_window(ctx, syms, S.ws, S.nloc, S.N, cfg->d_h, 1, cfg->d_pulse, cfg->ntaps, cfg->sps, M_PI / 4,
                                         S.tile_lo, S.ntiles, sig, S.out_origin, q_phase, q_phase, S.q_out_tile, stream))) return rc;
        const int64_t first_local = S.first + S.k_lo * cfg->sps - S.out_origin;
        if (cfg->fuse & 2) {
            rc = wf_awgn_mf_bank_dyn(ctx, sig, S.local_len, rot_re, rot_im, cfg->sigma, cfg->seed, cfg->stream_id,
                                     (uint64_t)S.out_origin, steady ? dyn + 1 : nullptr, cfg->d_mf_taps, cfg->mf_nfilt,
                                     cfg->mf_ntaps, first_local, cfg->sps, S.ncols, mf, stream,
                                     packed ? (int)(S.k_lo & 1) : -1);
        } else {
            if ((rc = wf_awgn_c128(ctx, sig, S.local_len, rot_re, rot_im, cfg->sigma, cfg->seed, cfg->stream_id,
                                   (uint64_t)S.out_origin, sig, stream))) return rc;
            rc = wf_mf_bank_c128(ctx, sig, S.local_len, cfg->d_mf_taps, cfg->mf_nfilt, cfg->mf_ntaps, first_local, cfg->sps, S.ncols,
                                 mf, stream);
        }
        if (rc) return rc;
    }
    if (steady && (phases & 4)) {   // the noise counter of the next chunk's part 4
        hipLaunchKernelGGL(stream_advance_kernel, dim3(1), dim3(64), 0, wf_stream(stream), dyn + 1,
                           (uint64_t)chunk_symbols * (uint64_t)cfg->sps);
        WF_LAUNCH_CHECK();
    }
    if (!(phases & 2)) return WF_OK;
    if (packed)
        rc = wf_viterbi4_detect_packed(ctx, mf, S.ncols, cfg->differential, link_warmup(cfg), dbits, dsyms, vit_state, stream);
    else
        rc = wf_viterbi4_detect(ctx, mf, S.ncols, cfg->differential, link_warmup(cfg), dbits, dsyms, vit_state, stream);
    if (rc) return rc;
    // decision of call k is compared with symbol k - length (examples/soqpsk_detection.py:201-209)
    const int64_t j0 = S.k_lo >= length ? 0 : length - S.k_lo;
    int64_t ncmp = S.ncols - j0;
    const int64_t sym0 = S.k_lo + j0 - length;            // global index of the first reference symbol
    if (ncmp > S.m_total - sym0) ncmp = S.m_total - sym0;
    if (ncmp > 0) {
        if ((rc = wf_count_errors(ctx, dsyms + j0, syms + (sym0 - S.ws), dbits + j0, bits + (sym0 - S.ws), ncmp, d_counts, stream))) return rc;
        if (h_compared) *h_compared = ncmp;
    }
    return WF_OK;
}
