/*
 * wfhip.h — C ABI of libwfhip.so: the MI355X (gfx950) kernels behind the
 * `waveforms.cpm` / `waveforms.filters` / `waveforms.viterbi` / `waveforms.glfsr` /
 * `waveforms.noise` Python API of mcdiarmid/waveforms.
 *
 * The reference has no FFI of its own (it is pure Python + NumPy); the boundary
 * is its Python module API.  Each entry point below replaces the NumPy / Python
 * loop at the cited reference location (paths relative to the reference repo)
 * and is bound from Python with ctypes (waveforms_amd/_hip.py; binding stub in
 * INTEGRATION.md).
 *
 * Conventions
 *  - plain pointers and sizes only; every `d_*` pointer is DEVICE memory on the
 *    context's GPU, every `h_*` pointer is host memory; `stream` is a
 *    hipStream_t passed as void* (NULL = the default stream);
 *  - all entry points are asynchronous on `stream` unless stated otherwise and
 *    allocate nothing (graph-capturable) — scratch lives in the wf_ctx;
 *  - complex128 arrays are interleaved (re, im) doubles, as numpy stores them;
 *  - return value: 0 on success, a negative wf_status otherwise;
 *    wf_last_error_string() describes the last failure on the calling thread.
 *    The Python layer maps WF_ERR_VALUE -> ValueError, WF_ERR_KEY -> KeyError
 *    (the exceptions the reference raises), anything else -> RuntimeError.
 */
#ifndef WFHIP_H
#define WFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    WF_OK = 0,
    WF_ERR_VALUE = -1,   /* bad argument (reference raises ValueError)          */
    WF_ERR_KEY = -2,     /* undefined table entry (reference raises KeyError)   */
    WF_ERR_HIP = -3,     /* a HIP runtime call failed                           */
    WF_ERR_DEVICE = -4,  /* a kernel reported a fault (e.g. scan hand-off timeout) */
    WF_ERR_NOMEM = -5
} wf_status;

typedef struct wf_ctx wf_ctx;

/* ---- library / context --------------------------------------------------- */
const char *wf_version(void);
const char *wf_last_error_string(void);

/* One context per GPU: owns the scan descriptors, LFSR jump tables and the
 * device fault word.  Synchronous (allocates).  `max_samples` sizes the scan
 * scratch (it grows on demand outside graph capture). */
int wf_ctx_create(int device, wf_ctx **out);
int wf_ctx_destroy(wf_ctx *ctx);
/* Retire a context without freeing it: stops the persistent per-symbol server (SOQPSKTrellisDetector.iteration,
 * waveforms/viterbi/algorithm.py:57-101, is served by one), drains the side stream of pipelined links and refuses to
 * start a new server afterwards.  The handle stays valid for wf_link_join / wf_viterbi4_iteration_quiesce /
 * wf_ctx_destroy — what the reference's objects do from finalisers after interpreter exit began
 * (examples/soqpsk_detection.py keeps its detector at module level).  Idempotent, synchronous. */
int wf_ctx_retire(wf_ctx *ctx);
/* Synchronises `stream`, returns WF_ERR_DEVICE if any kernel since the last
 * check raised the fault word (and clears it). */
int wf_ctx_check(wf_ctx *ctx, void *stream);
/* Per-context options: everything that tunes or instruments the library is a field of the context set through
 * this call — the library reads no environment variable and keeps no process-wide switch (the reference's objects
 * carry their own configuration the same way: SOQPSKTrellisDetector(length, differantial_encoding),
 * waveforms/viterbi/algorithm.py:19-42).  Values are int64; 0 is every option's default.  Unknown key or a value
 * outside the option's range: WF_ERR_VALUE.  Takes effect for calls issued afterwards on this context. */
typedef enum {
    WF_OPT_CPM_FORM = 0,          /* generic CPM detector: 0 choose by estimated time, 1 row form, 2 lane form (where compiled in) */
    WF_OPT_CPM_CHUNK_CALLS = 1,   /* calls per chunk of the generic CPM detector (any form); 0 = the library's choice */
    WF_OPT_DET_REPAIR = 2,        /* chunk-parallel detectors: 0 repair chunks whose proof failed (cascading: see
                                   * wf_viterbi_repaired), 1 only COUNT them (wf_viterbi4_unmerged) — tests of the proof */
    WF_OPT_DET_FINAL_VERIFY = 3,  /* 1: after the repairs compare every chunk boundary once more and count what still
                                   * differs in wf_viterbi4_unmerged (an internal-consistency check; always 0) */
    WF_OPT_ITERATION_SERVER = 4,  /* wf_viterbi4_iteration_host: 0 persistent server, 1 one launch + synchronise per call */
    WF_OPT_MCB_TAIL_PERMILLE = 5, /* one-kernel front end: resident-slot-fulls of tail tiles x 1000; 0 = default, -1 = none */
    WF_OPT_PIPE_RESERVE_CUS = 6,  /* pipelined SOQPSK link (wf_link_config.fuse bit 5), a measured alternative that is NOT the default
                                   * (profiles/r06_ab_prologue_ahead_cu_mask.log: 0.457 ms per block as shipped, 0.474 with -1, 0.50 with 8):
                                   * N >= 1: each block's prologue (PRBS + precoder, carry kernels) on a stream of its own beside the
                                   * previous block's front end, the front-end kernel on a stream whose CU mask leaves N compute units
                                   * free; -1: the same without a mask; 0: front end and prologue on the caller's stream.  Set before
                                   * the context's first pipelined block. */
    WF_OPT_CPM_SAMPLES_MIN_CALLS = 7, /* wf_cpm_viterbi_detect_samples / wf_cpm_link_config.fuse bit 7, 16-state lane form: shortest burst (calls) that
                                   * takes the samples form; 0 = the library's 6e6 (below it rows + the row form are faster: one lane per chunk leaves
                                   * most of a short burst's chunks to warm-up), otherwise >= 4096 — tests run the form on bursts an oracle can follow */
    WF_OPT_COUNT = 8
} wf_option;
int wf_ctx_set_option(wf_ctx *ctx, int key, int64_t value);
int wf_ctx_get_option(wf_ctx *ctx, int key, int64_t *value);
/* Two link options are promises about small device tables (wf_link_config.d_mf_factor; wf_cpm_link_config.fuse bit 6): the
 * library checks each on a host copy the first time it sees the table's ADDRESS on a context and remembers the verdict.  Whoever
 * frees or rewrites such a table calls this before the address can mean something else (the Python links do so when they are
 * created): every promise is then checked again on its next use.  (Reference: its objects own their tables,
 * examples/soqpsk_detection.py:134-173 builds them per run — nothing to invalidate there.) */
int wf_ctx_forget_promises(wf_ctx *ctx);

/* ---- K1: PRBS ------------------------------------------------------------
 * GLFSR.next_bit x n   (waveforms/glfsr/glfsr.py:6-19, pn.py:98-107).
 * Writes bits `skip .. skip+n-1` of the sequence started from `state` as one
 * u8 (0/1) per bit.  Leap-ahead by GF(2) matrix powers, bit-exact.
 * If h_state_out != NULL it receives the register after skip+n steps (host
 * arithmetic, available immediately).  `degree` in 2..64, mask as
 * generate_mask (pn.py:75-90) builds it. */
int wf_lfsr_generate(wf_ctx *ctx, int degree, uint64_t mask, uint64_t state, uint64_t skip,
                     uint8_t *d_bits, int64_t n, uint64_t *h_state_out, void *stream);

/* ---- K2: bits -> symbols ---------------------------------------------------
 * TrellisEncoder.encode (waveforms/cpm/trellis/encoder.py:17-48) for any
 * trellis of <= 16 states, <= 4 input bits per symbol.  Tables are HOST arrays,
 * dense [column][state][input]: h_next (u8) / h_out (i8), i.e. forward_map
 * (waveforms/cpm/trellis/model.py:127-137).  `i0`/`state0` are TrellisEncoder.i
 * / .state on entry; *h_state_out receives .state after the call (this one
 * value is copied back synchronously).  nbits % card != 0 -> WF_ERR_VALUE
 * (encoder.py:28-30). */
int wf_fsm_encode(wf_ctx *ctx, const uint8_t *h_next, const int8_t *h_out, int columns, int states,
                  int card, const uint8_t *d_bits, int64_t nbits, int64_t i0, int state0,
                  int8_t *d_symbols, int *h_state_out, void *stream);

/* Stateless-per-element mappers (a2'):
 *  kind 0: SOQPSKPrecoder.__call__  (waveforms/cpm/soqpsk/precoder.py:10-24),
 *          parity = precoder .i, mem0/mem1 = precoder .mem, out in {-1,0,1};
 *  kind 1: MultiHSymbolMapper.__call__ (waveforms/cpm/multih/precoder.py:9-23),
 *          parity = mapper .i AFTER its update, n must be even, out n/2 symbols;
 *  kind 2: PCMFMSymbolMapper.__call__ (waveforms/cpm/pcmfm/precoder.py:6-15). */
int wf_symbol_map(wf_ctx *ctx, int kind, const uint8_t *d_bits, int64_t n, int parity, int mem0,
                  int mem1, int8_t *d_symbols, void *stream);

/* ---- K3: zero-stuffed upsample + frequency-pulse FIR -----------------------
 * interpolated[sps:-1:sps] = symbols*h ; np.convolve(.., pulse, "same")
 * (waveforms/cpm/modulate.py:91-99).  Symbol m uses h[m % nh].  Output length
 * is wf_fir_out_len(nsym, sps, ntaps) = max((nsym+1)*sps, ntaps) — numpy's
 * "same" returns the longer operand's length. */
int64_t wf_fir_out_len(int64_t nsym, int sps, int ntaps);
int wf_upsample_fir_f64(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h,
                        int nh, const double *d_pulse, int ntaps, int sps, double *d_out,
                        void *stream);

/* ---- K4: phase accumulate (mod sps) + complex exponential ------------------
 * frequency_modulate (waveforms/cpm/modulate.py:28-54):
 *   revs_k = (revs_{k-1} + f_k) mod sps ; out_k = exp(j (revs_k 2pi/sps + phi0)).
 * Single-pass chained prefix scan; `revs_in` continues a previous chunk
 * (0 for a fresh call), *d_revs_out (may be NULL) receives the final revs. */
int wf_phase_cexp_f64(wf_ctx *ctx, const double *d_freq, int64_t n, int sps, double phi0,
                      double revs_in, double *d_out_ri, double *d_revs_out, void *stream);
/* Fused K3 + K4: cpm_modulate (waveforms/cpm/modulate.py:57-101) straight from symbols
 * to the complex baseband signal, one pass over HBM (1 B in, 16*sps B out per symbol);
 * tile carries come from a symbol-rate prefix sum instead of an inter-workgroup scan.
 * Same results as wf_upsample_fir_f64 followed by wf_phase_cexp_f64 (to rounding).
 * Returns 1 — not an error — when the configuration is outside the fused kernel's
 * envelope (signal shorter than the pulse, pulse longer than 33 symbols or than a tile, more than 8
 * modulation indices — modulate.py:91-92 cycles any number: symbol i takes d_h[i mod nh]);
 * the caller then runs the two stage kernels. */
int wf_cpm_modulate_c128(wf_ctx *ctx, const int8_t *d_symbols, int64_t nsym, const double *d_h, int nh,
                         const double *d_pulse, int ntaps, int sps, double phi0, double *d_out_ri,
                         void *stream);
/* phase_modulate (waveforms/cpm/modulate.py:12-25): out = exp(j * sens * phase). */
int wf_phase_modulate_f64(wf_ctx *ctx, const double *d_phase, int64_t n, double sens,
                          double *d_out_ri, void *stream);

/* ---- K5: AWGN ---------------------------------------------------------------
 * Device counterpart of generate_complex_awgn (waveforms/noise.py:8-32) and of
 * `signal * exp(-j pi/4) + noise` (examples/soqpsk_detection.py:85-89):
 *   out_k = in_k * (rot_re + j rot_im) + sigma * (n_re + j n_im)_k
 * with (n_re, n_im)_k = Box-Muller (two 32-bit uniforms) of half a Philox4x32-10 block:
 * absolute index a = first_index + k, counter = (a >> 1, stream_id), key = seed, words
 * (x0,x1) for even a, (x2,x3) for odd a.  d_in may be NULL (pure noise).  In-place allowed. */
int wf_awgn_c128(wf_ctx *ctx, const double *d_in_ri, int64_t n, double rot_re, double rot_im,
                 double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                 double *d_out_ri, void *stream);

/* The Box-Muller half of the source on caller-supplied words: sample k from
 * (d_words[2k], d_words[2k+1]) = (radius word xa, angle word xb) exactly as wf_awgn_c128
 * uses the halves of a Philox block.  |sample| <= sigma * sqrt(64 ln 2) = 6.66 sigma. */
int wf_box_muller32_c128(wf_ctx *ctx, const uint32_t *d_words, int64_t n, double sigma, double *d_out_ri,
                         void *stream);

/* ---- K6/K7: matched-filter bank ----------------------------------------------
 * nfilt complex FIRs, each np.convolve(r, taps[f], "same") sampled at
 * n = first + k*step, k < ncols   (examples/soqpsk_detection.py:141-156 PT bank,
 * :164-173 PAM bank with the pseudo-symbol weights folded into the taps,
 * :189-196 decimation).  d_taps is nfilt x ntaps complex128; output is
 * ncols x nfilt complex128 (one row per detector call).  step = 1, first = 0,
 * ncols = nsamp gives the full-rate bank.  Requires nsamp >= ntaps. */
int wf_mf_bank_c128(wf_ctx *ctx, const double *d_r_ri, int64_t nsamp, const double *d_taps_ri,
                    int nfilt, int ntaps, int64_t first, int step, int64_t ncols,
                    double *d_out_ri, void *stream);

/* Fused K5 + K6: the channel of wf_awgn_c128 applied while the matched-filter bank stages
 * its input (received samples never materialise in HBM).  Output identical to
 * wf_awgn_c128 followed by wf_mf_bank_c128 with the same noise coordinates. */
int wf_awgn_mf_bank_c128(wf_ctx *ctx, const double *d_signal_ri, int64_t nsamp, double rot_re, double rot_im,
                         double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                         const double *d_taps_ri, int nfilt, int ntaps, int64_t first, int step,
                         int64_t ncols, double *d_out_ri, void *stream);

/* ---- K8-K10: SOQPSK 4-state Viterbi detector ---------------------------------
 * SOQPSKTrellisDetector (waveforms/viterbi/algorithm.py:18-101) with
 * length = 2: for every row of d_mf (ncalls x 3 complex128, alpha = -2,0,+2)
 * element [0] of the bits / symbols arrays that .iteration() returns.
 * Chunk-parallel: each thread re-derives the path metrics over `warmup` rows
 * before its chunk; decisions equal the sequential detector's once survivors
 * have merged (warmup >= 32 recommended, 0 = library default).
 * d_state (may be NULL) is the detector state carried across calls for streaming:
 * [i, M0[4], inc_prev[8], pad[3]] + 16 doubles of staging (32 doubles,
 * zero-initialised = a fresh detector); NULL = a fresh detector, no carry-out. */
int wf_viterbi4_detect(wf_ctx *ctx, const double *d_mf_ri, int64_t ncalls, int differential,
                       int warmup, uint8_t *d_bits, int8_t *d_syms, double *d_state,
                       void *stream);
/* The same for ANY window `length` in 1 .. 64 (waveforms/viterbi/algorithm.py:19-42: `length` is a free
 * parameter; window loop :69-87, depth-`length` traceback :90-98).  Odd lengths included, literally: there the
 * reference pairs a row's increments (:57-63, section i % 2) with the branches of the other section (:69-87), and
 * at length 1 the stage updates its metrics column in place.  For every row, element [0] of what .iteration()
 * returns — the stage-0 branch of the path that ends in the first arg-min state after `length` - 1 stages of look-ahead.  Chunk-parallel with the
 * same on-device proof (wf_viterbi4_unmerged).  d_state (may be NULL): wf_viterbi4_window_state_bytes() bytes,
 * zero-initialised = a fresh detector, carried across calls of the same length. */
int wf_viterbi4_detect_window(wf_ctx *ctx, const double *d_mf_ri, int64_t ncalls, int length, int differential,
                              int warmup, uint8_t *d_bits, int8_t *d_syms, double *d_state, void *stream);
int64_t wf_viterbi4_window_state_bytes(void);
/* The batch detectors are chunk-parallel: every chunk re-derives the path metrics over `warmup` rows before
 * its own calls.  Each launch then PROVES on the device that the state a chunk started from is bitwise the state
 * the previous chunk ended with — the condition under which all decisions are those of the sequential
 * SOQPSKTrellisDetector (waveforms/viterbi/algorithm.py:44-101) — and REPAIRS what fails: the chunk's own calls
 * run again from the true state; a chunk whose end state changed hands it to the next chunk, which is run again in
 * turn, round after round until no boundary differs (worst case: the sequential detector).  So the result never
 * depends on `warmup`; the warm-up only sets how often the repair runs.
 * wf_viterbi4_unmerged: chunks left unproven since the last reset (synchronises `stream`).  With the default
 * options this is always 0; it counts only under WF_OPT_DET_REPAIR = 1 (repairs off) or when
 * WF_OPT_DET_FINAL_VERIFY finds an inconsistency. */
int wf_viterbi4_unmerged(wf_ctx *ctx, int64_t *h_count, int reset, void *stream);
/* *h_count = chunk repairs run since the last reset, every round counted (synchronises `stream`).
 * wf_viterbi_cascaded: of those, the repairs whose chunk ended in a different state than before and therefore
 * handed on to the next chunk.  (Build-defined like the proof itself: the reference's detector is one sequential
 * loop, algorithm.py:44-101.) */
int wf_viterbi_repaired(wf_ctx *ctx, int64_t *h_count, int reset, void *stream);
int wf_viterbi_cascaded(wf_ctx *ctx, int64_t *h_count, int reset, void *stream);

/* wf_viterbi4_detect + wf_count_errors in one call (fresh detector): decision k is
 * compared with reference element k - skip for 0 <= k - skip < ncompare
 * (examples/soqpsk_detection.py:201-209: skip = length); counts are ADDED to d_counts[0..1]. */
int wf_viterbi4_detect_count(wf_ctx *ctx, const double *d_mf_ri, int64_t ncalls, int differential, int warmup,
                             uint8_t *d_bits, int8_t *d_syms, const uint8_t *d_ref_bits,
                             const int8_t *d_ref_syms, int skip, int64_t ncompare, int64_t *d_counts,
                             void *stream);
/* One literal .iteration() for any window `length` <= 64 (algorithm.py:44-101),
 * detector state resident on the device (wf_viterbi4_state_bytes(length) bytes,
 * zero-initialised = a new detector).  d_mf3: 3 complex128.  Outputs: `length`
 * doubles each, as the reference returns. */
int64_t wf_viterbi4_state_bytes(int length);
int wf_viterbi4_iteration(wf_ctx *ctx, void *d_state, int length, int differential,
                          const double *d_mf3_ri, double *d_bits_out, double *d_syms_out,
                          void *stream);

/* The same call with HOST operands (h_mf3: 3 complex128; outputs `length` doubles each), for the
 * reference's per-symbol loop (examples/soqpsk_detection.py:189-198): one launch + one stream
 * synchronise through pinned, device-mapped staging owned by the context.  Synchronous.
 * `d_state` must be initialised (its zero fill complete, not merely enqueued) before the first call
 * made for it: the call only synchronises `stream` when `d_state` differs from the previous call's. */
int wf_viterbi4_iteration_host(wf_ctx *ctx, void *d_state, int length, int differential,
                               const double *h_mf3_ri, double *h_bits_out, double *h_syms_out, void *stream);
/* Diagnostic: device-side timing of the last request wf_viterbi4_iteration_host's persistent server answered, in
 * microseconds: {request read from host memory, cache check, the iteration itself, write-through + answer}. */
int wf_viterbi4_iteration_server_timing(wf_ctx *ctx, double *h_us4);
/* wf_viterbi4_iteration_host answers BEFORE the detector state has been written back to device memory (about a
 * microsecond later).  Call this before anything else reads, overwrites or frees a `d_state` that was passed to
 * it (the next wf_viterbi4_iteration_host call needs nothing: one server, in order).  No-op without a server.
 * (Reference: the state is the detector object's own arrays, waveforms/viterbi/algorithm.py:36-42.) */
int wf_viterbi4_iteration_quiesce(wf_ctx *ctx);
/* The detector's public state arrays as the reference keeps them on every instance (waveforms/viterbi/algorithm.py:25-42:
 * bi_history float64[8][length], metrics float64[4][length], path uint8[4][length]; :44 the call counter `i`), read from a
 * device-resident state of wf_viterbi4_iteration / _host into host arrays of those shapes (row-major).  Quiesces the per-symbol
 * server first; synchronous.  h_calls may be NULL. */
int wf_viterbi4_state_read(wf_ctx *ctx, const void *d_state, int length, int64_t *h_calls, double *h_bi_history,
                           double *h_metrics, uint8_t *h_path, void *stream);

/* ---- K11: error counting ------------------------------------------------------
 * examples/soqpsk_detection.py:200-209: number of j < m with
 * det_syms[j] != ref_syms[j] and with det_bits[j] != ref_bits[j]; the two counts
 * are ADDED to d_counts[0], d_counts[1] (int64; zero them first). */
int wf_count_errors(wf_ctx *ctx, const int8_t *d_det_syms, const int8_t *d_ref_syms,
                    const uint8_t *d_det_bits, const uint8_t *d_ref_bits, int64_t m,
                    int64_t *d_counts, void *stream);

/* ---- a5 helper: normalized time axis -----------------------------------------
 * np.linspace(0, N+1, (N+1)*sps, endpoint=False) (waveforms/cpm/modulate.py:81-88):
 * out[k] = k * step. */
int wf_time_axis_f64(wf_ctx *ctx, int64_t n, double step, double *d_out, void *stream);

/* ---- device-resident link (one Monte-Carlo trial block / one bench step) -------
 * PRBS -> TrellisEncoder(SOQPSKTrellis4x2[DiffEncoded]) -> cpm_modulate -> *exp(-j pi/4)
 * + AWGN -> matched-filter bank sampled at (n + timing_offset) % sps == 0 ->
 * SOQPSKTrellisDetector(length = 2) -> error count: the per-waveform body of
 * examples/soqpsk_detection.py:45-216, every stage one of the kernels above, all
 * intermediates in the caller's HBM workspace.  d_counts[0] += symbol errors,
 * d_counts[1] += bit errors; *h_compared (host, may be NULL) = number of symbols
 * compared (min_size of :204). */
typedef struct {
    int64_t nsym;           /* symbols (= bits) in the block                          */
    int sps;
    int degree;             /* PRBS register: degree, mask, start state, bits to skip */
    uint64_t mask, state, skip;
    int differential;       /* 1: SOQPSKTrellis4x2DiffEncoded, 0: SOQPSKTrellis4x2     */
    const double *d_h;      /* device: modulation index (1 double)                    */
    const double *d_pulse;  /* device: frequency pulse, ntaps doubles                 */
    int ntaps;
    const double *d_mf_taps; /* device: mf_nfilt x mf_ntaps complex128                */
    int mf_ntaps, mf_nfilt; /* mf_nfilt must be 3                                     */
    int timing_offset;      /* -1 for the PT bank, 0 for PAM-TG (:184-187)            */
    double sigma;           /* noise std-dev per real dimension                       */
    uint64_t seed, stream_id; /* Philox key / subsequence                             */
    int warmup;             /* Viterbi chunk warm-up, 0 = default                     */
    int fuse;               /* bit 0: fused modulator (wf_cpm_modulate_c128) instead   */
                            /* of the FIR + phase-scan stage kernels; bit 1: AWGN      */
                            /* inside the MF bank (wf_awgn_mf_bank_c128); bit 2 (with  */
                            /* bit 1, 3-filter bank, sps 8): the bank writes only the  */
                            /* 4 real components per call the 4-state detector reads   */
                            /* ({Re z1, Im z1, Re|Im z0, Im|Re z2}: 32 B rows, not 48);  */
                            /* bit 3 (with bits 0-2, 9-tap bank): modulator, channel and */
                            /* bank in ONE kernel — the baseband samples never reach HBM */
                            /* (rows bit-identical to bits 0-2; falls back to them when  */
                            /* the pulse is outside the kernel: > 9 symbols, sps != 8);  */
                            /* bit 4 (16): PRBS and precoder through the generic kernels */
                            /* (wf_lfsr_generate + the three-kernel wf_fsm_encode scan)   */
                            /* instead of the link's one-launch form (same bits and      */
                            /* symbols; bursts over 3.3e7 symbols take the generic form); */
                            /* bit 5 (32, with the one-kernel front end): the detector   */
                            /* and the error count of a block run on the context's side  */
                            /* stream and overlap the front end of the NEXT wf_link_run  */
                            /* on the same context; the workspace then holds two sets of */
                            /* intermediates (wf_link_workspace_bytes says so), used     */
                            /* alternately; counts complete after wf_link_join /         */
                            /* wf_ctx_check on the stream that reads them                 */
    int event_slot;         /* -1: off; 0..WF_LINK_EVENT_SLOTS-1: record HIP events    */
                            /* around every stage into that slot (wf_link_stage_ms)   */
    const double *d_mf_factor; /* device, optional (NULL: none): a long bank (the PAM    */
                            /* detector's, mf_ntaps != sps + 1) FACTORED as the reference */
                            /* computes it (examples/soqpsk_detection.py:158-173): two    */
                            /* real filters b_0, b_1 (mf_ntaps doubles each) and the 3 x 2 */
                            /* complex combination G (12 doubles, G[s][k] as re, im), with */
                            /* d_mf_taps[s] == sum_k G[s][k] b_k.  The one-kernel front end */
                            /* then runs two real filters instead of three complex ones    */
                            /* (a third fewer matrix instructions); every other path reads */
                            /* d_mf_taps.  The identity is CHECKED (host copy, 1e-12 of the */
                            /* largest tap) the first time these pointers are seen on a     */
                            /* context: a factorisation that does not reproduce the taps is */
                            /* WF_ERR_VALUE.  Do not rewrite the tables in place afterwards. */
} wf_link_config;
#define WF_LINK_EVENT_SLOTS 64
#define WF_LINK_STAGES 8    /* prbs, encode, fir, phase, awgn, mfbank, viterbi, count */
int64_t wf_link_workspace_bytes(const wf_link_config *cfg);
int wf_link_run(wf_ctx *ctx, const wf_link_config *cfg, void *d_workspace, int64_t workspace_bytes,
                int64_t *d_counts, int64_t *h_compared, void *stream);
/* Elapsed milliseconds of the WF_LINK_STAGES stages of the run that last used
 * `event_slot` (HIP events on the run's own stream).  Synchronises on that slot's
 * final event. */
int wf_link_stage_ms(wf_ctx *ctx, int event_slot, float *h_ms);

/* `stream` waits for the detectors and error counters that wf_link_run calls with fuse bit 5 left on the context's
 * side stream: call it (or wf_ctx_check, which includes it) on the stream that will read or reset the counters, or
 * reuse the workspace for something else.  No-op for a context that never ran a pipelined block.
 * (Reference: the loop body of examples/soqpsk_detection.py:45-216 is sequential; this is a scheduling call.) */
int wf_link_join(wf_ctx *ctx, void *stream);

/* Workspace offsets of a wf_link_run block, for callers that want the intermediates:
 * info8 = {ncols, one_kernel, off(detected bits), off(detected symbols), off(signal), row_bytes,
 *          signal samples, off(MF rows)} (offsets in bytes into the workspace).  one_kernel = 1: fuse = 15 runs
 *          modulator + channel + bank as ONE kernel for this configuration (3 x (sps + 1) bank at 8, 10 or 20
 *          samples per symbol, pulse of at most 9 symbols); row_bytes = 32 when the MF rows area holds the 4
 *          doubles per call the detector reads (fuse bit 2 in effect), else 16 * mf_nfilt. */
int wf_link_layout(const wf_link_config *cfg, int64_t *info8);

/* ---- streaming link (continuous stream in chunks) -------------------------------
 * The same chain over a stream of cfg->nsym symbols, `chunk_symbols` detector calls per
 * call, chunk_index = 0, 1, ... in order on one stream.  Neighbouring context is
 * re-generated as a halo (PRBS leap-ahead, counter-based noise, one modulator tile of
 * samples) or carried in `d_state` (WF_LINK_STREAM_STATE_BYTES, zero-initialised before
 * chunk 0: Viterbi state, encoder state, modulator phase).  Decisions and error counts
 * equal wf_link_run over the whole stream.  chunk_symbols must be a multiple of the
 * modulator tile (wf_mod_tile_geometry) and of 128.  Requires the fused modulator. */
#define WF_LINK_STREAM_STATE_BYTES 512
int wf_mod_tile_geometry(int sps, int ntaps, int64_t nsym_total, int64_t *tile_len, int64_t *sym_per_tile,
                         int64_t *ntiles_total);
int64_t wf_link_stream_workspace_bytes(const wf_link_config *cfg, int64_t chunk_symbols);
/* info8 = {calls in the chunk, first call index, off(detected bits), off(detected symbols),
 *          off(signal), global index of signal[0], signal samples resident, off(MF rows)} */
int wf_link_stream_layout(const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index, int64_t *info8);
int wf_link_stream_chunk(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                         void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                         int64_t *h_compared, void *stream);
/* Parts of a chunk, for callers that pipeline chunks on two streams with a workspace AND a wf_ctx
 * each (the parts of one chunk use the same pair).  `phases` is a bit set; each part needs the same
 * part of the previous chunk (its carry in d_state) and the earlier parts of its own chunk:
 *   bit 0 (1): PRBS, encoder, modulator carries          (carries: encoder state, modulator phase)
 *   bit 2 (4): modulator + channel + bank                 (with fuse bit 3: no carry of its own)
 *   bit 1 (2): detector + error count                     (carry: detector state)
 * phases = 7: the whole chunk = wf_link_stream_chunk. */
int wf_link_stream_chunk_phase(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                               void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                               int64_t *h_compared, int phases, void *stream);
/* Steady state for hipGraph replay: the launch sequence of an INTERIOR chunk with the only
 * two per-chunk quantities (PRBS position, noise counter) kept in d_state and advanced on the
 * device, so every call issues identical launches.  Use: chunk 0 with wf_link_stream_chunk,
 * capture ONE wf_link_stream_steady call (hipStreamBeginCapture / torch.cuda.graph) and
 * replay it once per interior chunk 1, 2, ... (wf_link_stream_interior tells which chunks
 * qualify; d_state's position words start at zero and every call ends by advancing them one
 * chunk), then finish the remaining chunk(s) with wf_link_stream_chunk.  Allocates and
 * synchronises nothing.  Same results as wf_link_stream_chunk. */
int wf_link_stream_interior(const wf_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index);
int wf_link_stream_steady(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, void *d_state,
                          void *d_workspace, int64_t workspace_bytes, int64_t *d_counts, int64_t *h_compared,
                          void *stream);
/* wf_link_stream_steady in the parts of wf_link_stream_chunk_phase (`phases` bits 0 / 2 / 1), each advancing its
 * own position word at its end: a PIPELINE of interior chunks on two streams (workspace and wf_ctx per stream,
 * one event per part) can be captured as ONE hipGraph and replayed. */
int wf_link_stream_steady_phase(wf_ctx *ctx, const wf_link_config *cfg, int64_t chunk_symbols, void *d_state,
                                void *d_workspace, int64_t workspace_bytes, int64_t *d_counts, int64_t *h_compared,
                                int phases, void *stream);

/* ---- generic CPM trellis detector (ARTM multi-h, PCM/FM) ---------------------------
 * The reference has NO detector for these waveforms — only their modulator side
 * (waveforms/cpm/multih/pulse_filters.py:11-23, precoder.py:9-23, waveforms/cpm/pcmfm/) and the
 * state-space theory (notes/cpm/cpm.md:52-140, N_S = p M^(L-1)).  These entry points implement the
 * detector DEFINED by oracle/cpm_oracle.c (sequential, build-defined): tilted-phase states of
 * notes/cpm/cpm.md:100-140, pulse-truncation matched filters in the manner of
 * examples/soqpsk_detection.py:134-156, and the conventions of waveforms/viterbi/algorithm.py:57-98
 * (increment Re(rotation * mf) minimised, strict '<', first arg-min, min-normalised metrics, one
 * decision per call from the best state).
 *
 * State = (phase class v mod NC, the Lp-1 previous symbols); NC < p carries the phase index per
 * survivor.  At most 16 states: NC * M^(Lp-1) <= 16; M in {2, 4}; nh in {1, 2}; Lp in 1..3. */
typedef struct {
    int M;          /* alphabet size: alpha = 2 U - (M - 1), U = 0 .. M-1                      */
    int p;          /* modulation indices K[i] / p, symbol m uses K[m % nh]                    */
    int nh;
    int K[2];
    int Lp;         /* symbols per matched filter (pulse truncation length)                    */
    int NC;         /* phase classes in the trellis state (divides p)                          */
    int D;          /* decision delay: call n decides symbol n - D + 1;  D * log2(M) <= 64     */
} wf_cpm_detector_config;

/* Matched-filter rows: row n, filter f = sum_k r[start0 + n*sps + k] * conj(T[n % nh][f][k]),
 * k < ntm.  d_templates: nh x nfilt x ntm complex128 (nfilt = M^Lp, index f = u_0 + M u_1 + ...),
 * d_rows: ncalls x nfilt complex128.  Samples outside [0, nsamp) count as zero. */
int wf_cpm_mf_rows_c128(wf_ctx *ctx, const double *d_r_ri, int64_t nsamp, const double *d_templates_ri, int nh,
                        int nfilt, int ntm, int64_t start0, int sps, int64_t ncalls, double *d_rows_ri,
                        void *stream);

/* The same rows from CLEAN samples with the channel of wf_awgn_c128 applied while staging
 * (derotation by rot, Philox noise of the given seed / stream / first index): identical to
 * wf_awgn_c128 followed by wf_cpm_mf_rows_c128, without the noisy samples ever being stored. */
int wf_cpm_awgn_mf_rows_c128(wf_ctx *ctx, const double *d_signal_ri, int64_t nsamp, double rot_re, double rot_im,
                             double sigma, uint64_t seed, uint64_t stream_id, uint64_t first_index,
                             const double *d_templates_ri, int nh, int nfilt, int ntm, int64_t start0, int sps,
                             int64_t ncalls, double *d_rows_ri, void *stream);

/* The detector over ncalls rows.  d_rot_cs: 2p pairs (cos, sin)(pi r / p) (caller-computed so that
 * oracle and device rotate with the same doubles).  d_decisions[k] (uint8) = the U decided at call
 * k, i.e. of symbol n0 + k - D + 1 (n0 = calls already made on d_state; entries with
 * n0 + k < D - 1 are written as 0).  Trellises of up to 256 states (N_S = NC M^(Lp-1), notes/cpm/cpm.md:128-140: up to 16
 * in one 16-lane group, 17 .. 64 — the 64-state ARTM design — one wave per detector, 65 .. 256 — the full ARTM trellis
 * p M^(L-1) = 256 with its 64 matched filters per symbol — one workgroup per detector; pulses of 2 or 3 symbols there).
 * Chunk-parallel like wf_viterbi4_detect: each 16-lane group
 * re-derives metrics, phase indices and decision registers over `warmup` rows (0 = default) and
 * every launch verifies bitwise that a chunk started from what its predecessor ended with
 * and repairs the chunks for which that failed, cascading into the following chunks where needed
 * (wf_viterbi4_unmerged, wf_viterbi_repaired): decisions are the sequential detector's whatever the warm-up.  d_state (WF_CPM_STATE_BYTES, zeroed = fresh detector,
 * may be NULL) carries the detector across calls. */
#define WF_CPM_STATE_BYTES 16384
/* Which of its forms wf_cpm_viterbi_detect runs for this trellis, burst length and warm-up: info4[0] = 3 the quad form
 * (65 .. 256 states: thread = state, one workgroup per chunk), 2 the wide form
 * (17 .. 64 states: lane = state, one wave per chunk), 0 the row
 * form (one 16-lane DPP row per chunk, any trellis of <= 16 states), 1 the lane form (one lane per chunk, trellis compiled
 * in: the ARTM 16-state and PCM/FM 10-state designs of waveforms/cpm/multih, waveforms/cpm/pcmfm; bursts long enough for
 * its 64 chunks per wave to fill the chip, ~9e6 / ~6.5e6 calls); info4[1] = its LDS ring depth; info4[2] = calls per
 * chunk; info4[3] = warm-up calls.  Both forms make the same decisions, bit for bit (notes/cpm/cpm.md:100-140 is what both
 * implement).  Priced for the context's device (its compute-unit count) under the context's options
 * (WF_OPT_CPM_FORM, WF_OPT_CPM_CHUNK_CALLS): exactly what wf_cpm_viterbi_detect will launch.  No device work. */
int wf_cpm_detector_form(wf_ctx *ctx, const wf_cpm_detector_config *det, int64_t ncalls, int warmup, int *info4);
int wf_cpm_viterbi_detect(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs,
                          const double *d_rows_ri, int64_t ncalls, int warmup, uint8_t *d_decisions,
                          void *d_state, void *stream);

/* wf_cpm_mf_rows_c128 + wf_cpm_viterbi_detect in ONE launch (round 6): the detector reads the noisy SAMPLES — call k's window is
 * samples start0 + 8 k .. start0 + 8 k + 8, 128 new bytes per call where a 16-filter row is 256 — and runs the matched filters
 * itself, in the manner of examples/soqpsk_detection.py:134-156 (pulse-truncation templates) for the trellis of
 * notes/cpm/cpm.md:100-140.  The templates must pair off as exact conjugates, d_templates[c][nfilt-1-f] == conj(d_templates[c][f])
 * (checked on a host copy, WF_ERR_VALUE if not): each pair is formed from four real 9-tap sums, so the filter outputs equal
 * wf_cpm_mf_rows_c128's to rounding (another order of additions) and are bit for bit those of the link's paired one-kernel front
 * end (wf_cpm_link_config.fuse bit 6).  Serves the 16-filter, 16-state ARTM design at 8 samples per symbol, 9-tap templates,
 * start0 >= 0, on bursts of at least 6e6 calls (shorter ones are faster through rows and the row form): returns 1 — nothing launched,
 * not an error — otherwise, and the caller runs wf_cpm_mf_rows_c128 + wf_cpm_viterbi_detect.  Call k takes template column k % nh.
 * Decisions, d_state, warm-up, proof and repair as wf_cpm_viterbi_detect (the repairs rebuild the rows they need from the samples). */
int wf_cpm_viterbi_detect_samples(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_templates_ri,
                                  int nh, int nfilt, int ntm, const double *d_samples_ri, int64_t nsamp, int64_t start0, int sps,
                                  int64_t ncalls, int warmup, uint8_t *d_decisions, void *d_state, void *stream);

/* Symbol and bit errors of decided U against transmitted symbols alpha (int8):
 * d_counts[0] += #(U != (alpha + M - 1)/2), d_counts[1] += popcount(U ^ (alpha + M - 1)/2)
 * (the reference's mappers are natural binary: waveforms/cpm/multih/precoder.py:22-23). */
int wf_cpm_count_errors(wf_ctx *ctx, const uint8_t *d_decided_u, const int8_t *d_ref_alpha, int M, int64_t m,
                        int64_t *d_counts, void *stream);

/* Device-resident link for these waveforms (one bench step / trial block):
 * PRBS -> mapper (wf_symbol_map kind) -> cpm_modulate -> *exp(-j pi/4) + AWGN -> matched-filter
 * rows -> detector -> error count over symbols [skip_head, ncalls - D].  Stage events as in
 * wf_link_run (slots: prbs, map, modulate, -, awgn, mfbank, viterbi, count). */
typedef struct {
    int64_t nsym;
    int sps;
    int degree;
    uint64_t mask, state, skip;
    int mapper_kind;          /* 1: MultiHSymbolMapper (2 bits/symbol), 2: PCMFMSymbolMapper       */
    wf_cpm_detector_config det;
    const double *d_h;        /* device: nh modulation indices K[i]/p as doubles                  */
    const double *d_pulse;    /* device: frequency pulse                                          */
    int ntaps;
    const double *d_templates; /* device: nh x M^Lp x (sps+1) complex128                           */
    const double *d_rot_cs;   /* device: 2p x (cos, sin)                                          */
    double sigma;
    uint64_t seed, stream_id;
    int warmup;
    int skip_head;            /* leading symbols excluded from the comparison (start transient)   */
    int event_slot;
    int fuse;                 /* bit 1: channel applied inside the matched-filter kernel;         */
                              /* bit 3 (with bit 1, sps 8, 4 or 16 filters): modulator + channel  */
                              /* + matched-filter rows in one kernel, no samples in HBM;          */
                              /* bit 4 (16): PRBS and mapper as two kernels (wf_lfsr_generate +   */
                              /* wf_symbol_map) instead of the link's one launch (same symbols);  */
                              /* bit 5 (32): as wf_link_config.fuse bit 5 — the detector and the  */
                              /* error count of a block on the context's side stream, beside the   */
                              /* next block's front end; two sets of intermediates in the         */
                              /* workspace; wf_link_join / wf_ctx_check before the counters are read; */
                              /* bit 6 (64, with bit 3; nf = 4 or 16 filters): the templates pair   */
                              /* off as exact conjugates, d_templates[c][nf-1-f] ==                */
                              /* conj(d_templates[c][f]) (a symmetric alphabet: the negated symbol */
                              /* pattern negates the phase) — CHECKED value for value on a host    */
                              /* copy the first time these pointers are seen on a context          */
                              /* (WF_ERR_VALUE if not so) — the one-kernel front end then forms     */
                              /* each pair from four real 9-tap sums (16 filters: 6 matrix          */
                              /* instructions per 16 symbols instead of 10; 4 filters: 18 multiply- */
                              /* adds per lane instead of 36); rows equal to rounding, not bitwise; */
                              /* bit 7 (128, with bits 1 and 6; 16 filters, sps 8, a burst the      */
                              /* detector's lane form takes): the front end stores the noisy        */
                              /* SAMPLES (128 B per symbol: cpm_modulate + the channel, one kernel) */
                              /* and the detector runs the matched filters itself                  */
                              /* (wf_cpm_viterbi_detect_samples): no rows in HBM, decisions bit for */
                              /* bit those of bits 1 + 3 + 6; ignored where it does not apply       */
                              /* (wf_cpm_link_form tells)                                          */
} wf_cpm_link_config;
int64_t wf_cpm_link_workspace_bytes(const wf_cpm_link_config *cfg);
int wf_cpm_link_run(wf_ctx *ctx, const wf_cpm_link_config *cfg, void *d_workspace, int64_t workspace_bytes,
                    int64_t *d_counts, int64_t *h_compared, void *stream);
/* info8 = {ncalls, start0, off(decisions), off(symbols alpha), off(signal), one_kernel (1: fuse bits 1 + 3 run modulator +
 * channel + filters as one kernel for this configuration), signal samples, off(rows)} */
int wf_cpm_link_layout(const wf_cpm_link_config *cfg, int64_t *info8);
/* What wf_cpm_link_run launches for this configuration on this context (its options and device): info4[0] = front end — 0:
 * modulator, channel and matched filters as separate kernels, 1: one kernel with rows out (fuse bits 1 + 3), 2: modulator +
 * channel in one kernel with SAMPLES out and the matched filters inside the detector (fuse bit 7) —, info4[1 .. 3] = the
 * detector's form, calls per chunk and warm-up calls as wf_cpm_detector_form reports them.  No device work. */
int wf_cpm_link_form(wf_ctx *ctx, const wf_cpm_link_config *cfg, int *info4);

/* Streaming form of the CPM link (the scheme of wf_link_stream_chunk for the waveforms of BASELINE configs[2]):
 * a stream of cfg->nsym symbols in chunks of chunk_symbols detector calls, HBM footprint of one chunk; chunk c
 * makes calls [c*B, (c+1)*B).  Neighbouring context is re-generated as a halo (PRBS leap-ahead, memoryless
 * mapper, counter-based noise, one modulator tile either side) or carried in d_state
 * (WF_CPM_STREAM_STATE_BYTES, zero-initialised: detector state + modulator phase carry).  Chunks must be
 * processed in order; decisions and counts equal wf_cpm_link_run over the whole stream.  chunk_symbols: a
 * multiple of one modulator tile (wf_mod_tile_geometry) and of 128, at least 4 halos; the configuration must be
 * one the one-kernel front end takes (fuse bits 1 + 3; wf_cpm_link_layout info8[5]).
 * wf_cpm_link_stream_layout: info8 = {calls in the chunk, first call index, off(decisions), off(symbols alpha),
 * global index of symbols[0], calls of the whole stream, symbols per modulator tile, off(rows)}. */
#define WF_CPM_STREAM_STATE_BYTES 20480
int64_t wf_cpm_link_stream_workspace_bytes(const wf_cpm_link_config *cfg, int64_t chunk_symbols);
int wf_cpm_link_stream_layout(const wf_cpm_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index, int64_t *info8);
int wf_cpm_link_stream_chunk(wf_ctx *ctx, const wf_cpm_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                             void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                             int64_t *h_compared, void *stream);
/* The same chunk in parts, for callers that pipeline consecutive chunks on two HIP streams (own workspace and own
 * wf_ctx per stream; wf_link_stream_chunk_phase's numbering): phases bit 0 = PRBS + mapper (depends on nothing),
 * bit 2 = modulator + channel + matched-filter rows (after this chunk's bit-0 part and the previous chunk's bit-2 part:
 * the phase carry in d_state), bit 1 = detector + error count (after this chunk's bit-2 part and the previous chunk's
 * bit-1 part: the detector carry).  7 = wf_cpm_link_stream_chunk.  Replaces, for a stream, the per-call chain of
 * waveforms/cpm/modulate.py:57-101 + waveforms/noise.py:8-32 + the build-defined detector of wf_cpm_viterbi_detect. */
int wf_cpm_link_stream_chunk_phase(wf_ctx *ctx, const wf_cpm_link_config *cfg, int64_t chunk_symbols, int64_t chunk_index,
                                   void *d_state, void *d_workspace, int64_t workspace_bytes, int64_t *d_counts,
                                   int64_t *h_compared, int phases, void *stream);

/* ---- data products of the reference's plotting helpers (no plotting) ------------------
 * Welch PSD exactly as Axes.psd / matplotlib.mlab.psd evaluates the call of
 * waveforms/viz/psd.py:36-41 (window d_window of nfft doubles, np.hanning for the reference;
 * no overlap, no detrend, two-sided, not scaled by frequency): d_pxx[nfft] in fftshift order =
 * mean over the n // nfft segments of |FFT(window * scale * x)|^2 / wsum^2 (wsum = sum |window|).
 * nfft a power of two <= 4096; d_scratch holds wf_welch_scratch_doubles(n, nfft) doubles. */
int64_t wf_welch_scratch_doubles(int64_t n, int nfft);
int wf_welch_psd_c128(wf_ctx *ctx, const double *d_x_ri, int64_t n, int nfft, double scale, const double *d_window,
                      double wsum, double *d_scratch, double *d_pxx, void *stream);
/* Phase-tree traces (waveforms/viz/tree.py:64-70): per chunk of sps*modulo samples np.unwrap(np.angle)
 * minus `off`, or minus the chunk's first phase when use_first != 0.  d_out: (n / len) x len. */
int wf_phase_tree_f64(wf_ctx *ctx, const double *d_x_ri, int64_t n, int sps, int modulo, int use_first, double off,
                      double *d_out, void *stream);
/* Eye-diagram traces (waveforms/viz/eye.py:40-55): (n-1)/len traces of len+1 points, len = sps*modulo:
 * time axis (time - time[start]) + t_offset, real and imaginary planes.  BOTH d_time and d_x_ri must hold n elements. */
int wf_eye_traces_c128(wf_ctx *ctx, const double *d_time, const double *d_x_ri, int64_t n, int sps, int modulo,
                       double t_offset, double *d_t_out, double *d_re_out, double *d_im_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WFHIP_H */
