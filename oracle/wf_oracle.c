/*
 * wf_oracle.c — CPU restatement of the sequential loops of mcdiarmid/waveforms'
 * CPM modulate -> AWGN -> matched-filter -> Viterbi-detect path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under waveforms_amd/ may link, load or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and there only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  Every function below is checked against outputs of the
 * reference itself (tests/golden/*.npz, produced by tests/golden/make_golden.py
 * importing /root/reference in the build container) in tests/test_oracle_golden.py.
 * The one exception is orc_philox_awgn: the reference draws noise from numpy's
 * PCG64 + ziggurat (waveforms/noise.py:24-32), which has no parallel form; the
 * device generator is Philox4x32-10 + Box-Muller, a build-defined spec pinned by
 * the published Random123 known-answer vectors instead.
 *
 * All file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------ K1 ---- */
/* waveforms/glfsr/glfsr.py:15-19 — Galois right-shift LFSR, one bit per step. */
void orc_lfsr_generate(uint64_t mask, uint64_t *state_io, uint8_t *bits, int64_t n)
{
    uint64_t s = *state_io;
    for (int64_t k = 0; k < n; ++k) {
        uint8_t bit = (uint8_t)(s & 1u);
        s >>= 1;
        if (bit) s ^= mask;
        bits[k] = bit;
    }
    *state_io = s;
}

/* ------------------------------------------------------------------ K2 ---- */
/* waveforms/cpm/trellis/encoder.py:33-46 — FSM walk.  Tables are dense:
 *   next_tab[(col*states + st)*ninp + inp], out_tab[same index]
 * (built by the caller from the Branch lists exactly like forward_map,
 * waveforms/cpm/trellis/model.py:127-137).  card bits are consumed MSB-first
 * (encoder.py:35-40).  Returns -1 if nbits is not a multiple of card
 * (encoder.py:28-30 raises ValueError). */
int orc_fsm_encode(const uint8_t *next_tab, const int8_t *out_tab, int columns, int states,
                   int card, const uint8_t *bits, int64_t nbits, int8_t *symbols,
                   int64_t *i_io, int32_t *state_io)
{
    if (nbits % card) return -1;
    const int ninp = 1 << card;
    int64_t i = *i_io;
    int st = *state_io;
    const int64_t nsym = nbits / card;
    for (int64_t n = 0; n < nsym; ++n) {
        int inp = 0;
        for (int x = 0; x < card; ++x) inp |= (bits[n * card + x] & 1) << (card - x - 1);
        const int col = (int)(i % columns);
        const int idx = (col * states + st) * ninp + inp;
        symbols[n] = out_tab[idx];
        st = next_tab[idx];
        ++i;
    }
    *i_io = i;
    *state_io = st;
    return 0;
}

/* ------------------------------------------------------------------ K4 ---- */
/* Python/numpy float `%` with a positive modulus (what `(revs + sample) % sps`
 * evaluates, waveforms/cpm/modulate.py:52). */
static inline double py_mod(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}

/* waveforms/cpm/modulate.py:48-54 — sequential accumulate-with-modulo, then
 * exp(1j * phase).  out_ri is interleaved (re, im). */
void orc_frequency_modulate(const double *freq_pulses, int64_t n, int sps, double initial_phase,
                            double *out_ri)
{
    const double sensitivity = 2 * M_PI / sps;
    double revs = 0.0;
    for (int64_t k = 0; k < n; ++k) {
        revs = py_mod(revs + freq_pulses[k], (double)sps);
        const double ph = revs * sensitivity + initial_phase;
        out_ri[2 * k] = cos(ph);
        out_ri[2 * k + 1] = sin(ph);
    }
}

/* --------------------------------------------------------------- K8-K10 --- */
/* waveforms/viterbi/algorithm.py:18-101 restated literally (window of `length`
 * stages recomputed on every call, in-place array semantics preserved).
 *
 * Trellis description (one entry per branch, column-major: br_*[col*bpc + b]):
 *   br_inp, br_out_idx (index into the sorted symbol alphabet,
 *   model.py:175-176), br_out (the symbol itself), br_start, br_end.
 * mf is n_calls x 3 complex128, interleaved (re, im), rows ordered like
 * fsm.symbol_idx_map (alpha = -2, 0, +2).
 *
 * Detector state (algorithm.py:25-42) is carried in *st so the caller may
 * stream.  Outputs: for every call element [0] of the two returned arrays
 * (what examples/soqpsk_detection.py:196-198 keeps); if full_bits/full_syms are
 * non-NULL they receive all `length` elements per call. */
#define ORC_MAX_LEN 64
#define ORC_MAX_STATES 16
#define ORC_MAX_BPC 64

typedef struct {
    int64_t i;
    double bi_history[ORC_MAX_BPC][ORC_MAX_LEN];
    double metrics[ORC_MAX_STATES][ORC_MAX_LEN];
    uint8_t path[ORC_MAX_STATES][ORC_MAX_LEN];
} orc_viterbi_state;

int orc_viterbi_state_size(void) { return (int)sizeof(orc_viterbi_state); }

void orc_viterbi_reset(orc_viterbi_state *st) { memset(st, 0, sizeof(*st)); }

static inline int pos_mod(int64_t a, int m)
{
    int64_t r = a % m;
    return (int)(r < 0 ? r + m : r);
}

int orc_viterbi_run(orc_viterbi_state *st, int columns, int states, int bpc, const int8_t *br_inp,
                    const int8_t *br_out, const uint8_t *br_out_idx, const uint8_t *br_start,
                    const uint8_t *br_end, int length, const double *mf, int64_t n_calls,
                    double *bits0, double *syms0, double *full_bits, double *full_syms)
{
    if (length < 1 || length > ORC_MAX_LEN || states > ORC_MAX_STATES || bpc > ORC_MAX_BPC)
        return -1;
    const int L = length;
    for (int64_t call = 0; call < n_calls; ++call) {
        const double *z = mf + call * 6;
        /* algorithm.py:57-63 — shift history, append the new branch increments.
         * state_exp_term = [+1j, -1, +1, -1j] (algorithm.py:30): the real part of
         * the product is a signed selection of re/im. */
        for (int b = 0; b < bpc; ++b) {
            double first = st->bi_history[b][0];
            for (int j = 0; j + 1 < L; ++j) st->bi_history[b][j] = st->bi_history[b][j + 1];
            st->bi_history[b][L - 1] = first;
        }
        const int col_now = pos_mod(st->i, columns);
        for (int b = 0; b < bpc; ++b) {
            const int k = col_now * bpc + b;
            const double re = z[2 * br_out_idx[k]], im = z[2 * br_out_idx[k] + 1];
            double inc;
            switch (br_start[k] & 3) {
            case 0: inc = 0.0 * re - 1.0 * im; break;   /* (+1j) * z */
            case 1: inc = -re; break;                  /* -1 * z    */
            case 2: inc = re; break;                   /* +1 * z    */
            default: inc = -0.0 * re - (-1.0) * im; break; /* (-1j) * z */
            }
            st->bi_history[b][L - 1] = inc;
        }
        /* algorithm.py:65-67 */
        double mn = st->metrics[0][0];
        for (int s = 1; s < states; ++s)
            if (st->metrics[s][0] < mn) mn = st->metrics[s][0];
        double carried[ORC_MAX_STATES];
        for (int s = 0; s < states; ++s) carried[s] = st->metrics[s][0] - mn;
        for (int s = 0; s < states; ++s) st->metrics[s][L - 1] = carried[s];
        for (int s = 0; s < states; ++s)
            for (int j = 0; j + 1 < L; ++j) st->metrics[s][j] = 0.0;
        memset(st->path, 0, sizeof(st->path));
        /* algorithm.py:69-87 — add-compare-select, strict '<' */
        for (int j = 0; j < L; ++j) {
            const int col = pos_mod(st->i + j - 1, columns);
            const int jm1 = pos_mod(j - 1, L);
            for (int s = 0; s < states; ++s) {
                int min_k = 0;
                double min_m = INFINITY;
                for (int b = 0; b < bpc; ++b) {
                    const int k = col * bpc + b;
                    if (br_end[k] != s) continue;
                    const double mm = st->metrics[br_start[k]][jm1] + st->bi_history[b][j];
                    if (mm < min_m) {
                        min_m = mm;
                        min_k = br_start[k];
                    }
                }
                st->metrics[s][j] = min_m;
                st->path[s][j] = (uint8_t)min_k;
            }
        }
        /* algorithm.py:90-98 — traceback from the first arg-min state */
        int state = 0;
        for (int s = 1; s < states; ++s)
            if (st->metrics[s][L - 1] < st->metrics[state][L - 1]) state = s;
        double ob[ORC_MAX_LEN], os[ORC_MAX_LEN];
        for (int j = L - 1; j >= 0; --j) {
            const int col = pos_mod(st->i + j - 1, columns);
            const int pred = st->path[state][j];
            int found = -1;
            for (int b = 0; b < bpc; ++b) { /* reverse_transitions: last match wins */
                const int k = col * bpc + b;
                if (br_end[k] == state && br_start[k] == pred) found = k;
            }
            if (found < 0) return -2; /* KeyError in the reference */
            ob[j] = br_inp[found];
            os[j] = br_out[found];
            state = pred;
        }
        bits0[call] = ob[0];
        syms0[call] = os[0];
        if (full_bits)
            for (int j = 0; j < L; ++j) full_bits[call * L + j] = ob[j];
        if (full_syms)
            for (int j = 0; j < L; ++j) full_syms[call * L + j] = os[j];
        st->i += 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ K5 ---- */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14 constants).  Build-defined
 * device noise source — see the header comment. */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof(c));
}

/* One Philox block per PAIR of complex samples: absolute sample index a -> counter =
 * (a >> 1 lo, a >> 1 hi, stream lo, stream hi), key = (seed lo, seed hi); the even sample
 * uses words (x0, x1), the odd one (x2, x3); u1 = (xa + 1) 2^-32 in (0,1], u2 = xb 2^-32 in
 * [0,1); Box-Muller.  out_ri interleaved (re, im).  If `signal_ri` is non-NULL the noise is
 * added to it (the `modulated + noise` of examples/soqpsk_detection.py:89). */
void orc_philox_awgn(double sigma, uint64_t seed, uint64_t stream, uint64_t first_index,
                     int64_t n, const double *signal_ri, double *out_ri)
{
    for (int64_t k = 0; k < n; ++k) {
        const uint64_t idx = first_index + (uint64_t)k;
        const uint64_t pair = idx >> 1;
        uint32_t c[4] = {(uint32_t)pair, (uint32_t)(pair >> 32), (uint32_t)stream,
                         (uint32_t)(stream >> 32)};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t xa = (idx & 1) ? c[2] : c[0], xb = (idx & 1) ? c[3] : c[1];
        const double u1 = ((double)xa + 1.0) * 0x1.0p-32;
        const double u2 = (double)xb * 0x1.0p-32;
        const double r = sigma * sqrt(-2.0 * log(u1));
        const double th = 2.0 * M_PI * u2;
        double re = r * cos(th), im = r * sin(th);
        if (signal_ri) {
            re += signal_ri[2 * k];
            im += signal_ri[2 * k + 1];
        }
        out_ri[2 * k] = re;
        out_ri[2 * k + 1] = im;
    }
}

/* The Box-Muller step of orc_philox_awgn on caller-supplied words (2 per sample). */
void orc_box_muller32(const uint32_t *words, int64_t n, double sigma, double *out_ri)
{
    for (int64_t k = 0; k < n; ++k) {
        const double u1 = ((double)words[2 * k] + 1.0) * 0x1.0p-32;
        const double u2 = (double)words[2 * k + 1] * 0x1.0p-32;
        const double r = sigma * sqrt(-2.0 * log(u1));
        const double th = 2.0 * M_PI * u2;
        out_ri[2 * k] = r * cos(th);
        out_ri[2 * k + 1] = r * sin(th);
    }
}

/* ------------------------------------------------------------------ K3 ---- */
/* waveforms/cpm/modulate.py:95-99 — zero-stuffed upsample (impulses at
 * sps, 2*sps, ..., N*sps) convolved with the pulse, mode="same"
 * (= full[(M-1)//2 : (M-1)//2 + len]).  Direct sum over the <= ceil(M/sps)
 * non-zero terms; the numpy call it restates adds the same products (the
 * zero-stuffed terms contribute exact zeros) in a different order, so agreement
 * is to rounding (<= 1e-15 abs), not bitwise.  Used only as the timed CPU
 * baseline for long inputs; the parity oracle calls np.convolve itself. */
void orc_upsample_fir(const int8_t *symbols, int64_t nsym, const double *h, int nh,
                      const double *g, int M, int sps, double *out)
{
    const int64_t npts = (nsym + 1) * (int64_t)sps;
    /* np.convolve swaps its operands when the second is longer, so "same" yields
     * max(npts, M) samples starting at full[(min(npts, M) - 1) / 2]. */
    const int64_t out_len = npts >= M ? npts : M;
    const int64_t c = ((npts >= M ? M : npts) - 1) / 2;
    for (int64_t n = 0; n < out_len; ++n) {
        const int64_t top = (n + c) / sps; /* largest m+1 with tap index >= 0 */
        const int r = (int)((n + c) % sps);
        double acc = 0.0;
        for (int k = r; k < M; k += sps) {
            const int64_t mp1 = top - (k - r) / sps;
            if (mp1 < 1) break;
            if (mp1 > nsym) continue;
            /* interpolated[sps:-1:sps]: impulses at sps, 2*sps, ..., N*sps (modulate.py:96) */
            acc += (double)symbols[mp1 - 1] * h[(mp1 - 1) % nh] * g[k];
        }
        out[n] = acc;
    }
}

/* examples/soqpsk_detection.py:141-156 restricted to the samples the detector
 * consumes (:189-196): out[k] = conv_same(r, taps)[first + k*sps], complex taps,
 * for k = 0..ncols-1.  taps_ri is nfilt x ntap complex, out is ncols x nfilt
 * complex (row per symbol).  CPU-baseline helper, checked against np.convolve
 * in tests. */
void orc_mf_bank_decim(const double *r_ri, int64_t nsamp, const double *taps_ri, int nfilt,
                       int ntap, int64_t first, int sps, int64_t ncols, double *out_ri)
{
    const int c = (ntap - 1) / 2;
    for (int64_t k = 0; k < ncols; ++k) {
        const int64_t n = first + k * sps;
        for (int f = 0; f < nfilt; ++f) {
            double ar = 0.0, ai = 0.0;
            for (int t = 0; t < ntap; ++t) {
                const int64_t idx = n + c - t;
                if (idx < 0 || idx >= nsamp) continue;
                const double xr = r_ri[2 * idx], xi = r_ri[2 * idx + 1];
                const double tr = taps_ri[2 * (f * ntap + t)], ti = taps_ri[2 * (f * ntap + t) + 1];
                ar += xr * tr - xi * ti;
                ai += xr * ti + xi * tr;
            }
            out_ri[2 * (k * nfilt + f)] = ar;
            out_ri[2 * (k * nfilt + f) + 1] = ai;
        }
    }
}
