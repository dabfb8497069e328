/*
 * cpm_oracle.c — sequential CPU statement of the generic CPM trellis detector
 * (ARTM multi-h 16-state, PCM/FM).
 *
 * TEST INFRASTRUCTURE ONLY (same rule as wf_oracle.c): only tests/, smoke() and bench.py's
 * cpu_baseline leg may load this.
 *
 * Parity status: UNPINNED / BUILD-DEFINED.  mcdiarmid/waveforms has no detector for these
 * waveforms — only their modulator side (waveforms/cpm/multih/pulse_filters.py:11-23,
 * precoder.py:9-23, waveforms/cpm/pcmfm/) and the state-space theory
 * (notes/cpm/cpm.md:52-140).  This file therefore DEFINES the detector the HIP kernels must
 * reproduce bit for bit; what ties it to the reference is (a) the signal model — the modulator
 * whose output it detects is pinned by the reference's goldens — (b) the phase-state
 * decomposition of notes/cpm/cpm.md:100-140 (tilted phase: U = (alpha + M-1)/2, phase-state
 * index I = sum U_i K_i mod p, data-independent phase tilt), and (c) the conventions of the
 * reference's one detector, waveforms/viterbi/algorithm.py:57-98: branch increment
 * Re(rotation[start] * mf[index]) MINIMISED, strict '<' so the first listed branch wins ties,
 * first arg-min end state, carried metrics normalised by their minimum, one decision per call
 * by traceback from the current best state.  The matched filters are the reference's
 * pulse-truncation idea (examples/soqpsk_detection.py:134-156: sps+1 samples of the phase
 * pulse around its centre), generalised to L' symbols.
 *
 * RULE (round 4): because this detector is build-defined, a statement in THIS file may be ordered the way a kernel
 * computes it (round 3 did so once: orc_cpm_mf_rows accumulates in the matrix-pipe instruction's order).  That
 * licence ends here: nothing in wf_oracle.c / numpy_ref.py / viz_ref.py — the statements pinned by the reference's
 * goldens, rows a1-a11 and f4 — may ever follow a kernel.  tests/test_oracle_golden.py::
 * test_golden_pinned_oracle_outputs_are_frozen fails when any of them changes its output on the goldens' inputs.
 *
 * All file:line citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int M;        /* alphabet size (2 or 4): alpha = 2 U - (M - 1)                         */
    int lgM;      /* log2 M                                                                */
    int p;        /* modulation indices are K[i] / p                                       */
    int nh;       /* number of modulation indices (cycled per symbol, h_0 first:           */
                  /*   waveforms/cpm/modulate.py:91-92)                                    */
    int K[8];
    int Lp;       /* symbols per matched filter (pulse truncation length L'), 1..3         */
    int NC;       /* phase classes kept in the state (divides p); NC < p = per-survivor    */
                  /*   phase (decision feedback), NC == p = every phase state in the trellis */
    int D;        /* decision delay: symbol n - D + 1 is decided at call n; D * lgM <= 64  */
} orc_cpm_cfg;

static inline int imod(int64_t a, int m) { int r = (int)(a % m); return r < 0 ? r + m : r; }

int orc_cpm_nstates(const orc_cpm_cfg *c)
{
    int ncorr = 1;
    for (int i = 1; i < c->Lp; ++i) ncorr *= c->M;
    return c->NC * ncorr;
}

/* Sum of K over symbols 0 .. m-1 (mod 2p): the phase tilt accumulated by the symbols that have
 * left the correlative window (notes/cpm/cpm.md:108-112). */
static int ksum_mod(const orc_cpm_cfg *c, int64_t m)
{
    if (m <= 0) return 0;
    int per = 0;
    for (int i = 0; i < c->nh; ++i) per += c->K[i];
    int64_t acc = (m / c->nh) % (2 * c->p) * per;
    for (int i = 0; i < (int)(m % c->nh); ++i) acc += c->K[i];
    return (int)(acc % (2 * c->p));
}

/* Matched-filter rows.  taps: [nh][NF][ntm] complex (interleaved), the TEMPLATES (not reversed,
 * not conjugated); row n = correlation of r[start0 + n*sps + k], k < ntm, with the templates of
 * column n % nh:  Z = sum_k r[k] * conj(T[k]).   out: [ncalls][NF] complex. */
void orc_cpm_mf_rows(const double *r_ri, int64_t npts, const double *taps_ri, int nh, int NF, int ntm,
                     int64_t start0, int sps, int64_t ncalls, double *out_ri)
{
    for (int64_t n = 0; n < ncalls; ++n) {
        const double *T = taps_ri + (size_t)2 * ((size_t)(n % nh) * NF) * ntm;
        const int64_t s = start0 + n * sps;
        for (int f = 0; f < NF; ++f) {
            double zr = 0.0, zi = 0.0;
            for (int k = 0; k < ntm; ++k) {
                const int64_t t = s + k;
                if (t < 0 || t >= npts) continue;
                const double rr = r_ri[2 * t], ri = r_ri[2 * t + 1];
                const double tr = T[2 * (f * ntm + k)], ti = T[2 * (f * ntm + k) + 1];
                /* per tap: the imaginary sample's term, then the real one's — in BOTH sums.  It is the order in
                 * which v_mfma_f64_16x16x4_f64 runs its k index (a bitwise fma chain, k ascending: measured,
                 * tools/mfma_f64_probe.hip) over the operand order (Im r_0, Re r_0, Im r_1, ...) the ARTM front end
                 * feeds it, so the matrix-core kernel and the vector-pipe kernels produce the same bits. */
                zr = fma(rr, tr, fma(ri, ti, zr));
                zi = fma(-rr, ti, fma(ri, tr, zi));
            }
            out_ri[2 * (n * NF + f)] = zr;
            out_ri[2 * (n * NF + f) + 1] = zi;
        }
    }
}

/* Detector state carried across calls (streaming, and the unit the chunk-parallel kernel must
 * re-derive): call counter, metric / phase index / decision register per state. */
#define ORC_CPM_MAX_STATES 256
typedef struct {
    int64_t n;
    double metric[ORC_CPM_MAX_STATES];
    int32_t v[ORC_CPM_MAX_STATES];
    uint64_t hist[ORC_CPM_MAX_STATES];
} orc_cpm_state;

int orc_cpm_state_size(void) { return (int)sizeof(orc_cpm_state); }

void orc_cpm_state_init(const orc_cpm_cfg *c, orc_cpm_state *st)
{
    memset(st, 0, sizeof *st);
    const int S = orc_cpm_nstates(c);
    for (int s = 0; s < S; ++s) st->v[s] = s % c->NC;   /* state index = class + NC * corr */
}

/* rot_cs: 2p pairs (cos, sin)(pi r / p), r = 0 .. 2p-1 (computed once by the caller and shared
 * with the device so both sides rotate with the same doubles).
 * rows: [ncalls][NF] complex, NF = M^Lp, filter index f = u_0 + M * corr.
 * out_syms[n - D + 1] = decided U (0 .. M-1) for n >= D - 1; returns 0, or -1 on a bad config. */
int orc_cpm_viterbi(const orc_cpm_cfg *c, const double *rot_cs, const double *rows_ri, int64_t ncalls,
                    orc_cpm_state *st, uint8_t *out_syms)
{
    const int M = c->M, Lp = c->Lp, NC = c->NC, p = c->p, D = c->D;
    const int S = orc_cpm_nstates(c);
    if (S > ORC_CPM_MAX_STATES || D * c->lgM > 64 || D < 1 || Lp < 1 || Lp > 3 || p % NC) return -1;
    int NF = 1;
    for (int i = 0; i < Lp; ++i) NF *= M;
    int msub = 1;                       /* M^(Lp-2): weight of the oldest symbol inside corr */
    for (int i = 2; i < Lp; ++i) msub *= M;
    double nm[ORC_CPM_MAX_STATES];
    int32_t nv[ORC_CPM_MAX_STATES];
    uint64_t nh_[ORC_CPM_MAX_STATES];
    for (int64_t k = 0; k < ncalls; ++k) {
        const int64_t n = st->n;
        const double *Z = rows_ri + (size_t)2 * k * NF;
        const int64_t m_old = n - Lp + 1;                       /* symbol leaving the window after this call */
        const int K_old = m_old >= 0 ? c->K[m_old % c->nh] : 0;   /* virtual pre-start symbols carry no phase */
        const int tilt = imod((int64_t)(M - 1) * ksum_mod(c, n - Lp + 1), 2 * p);   /* symbols 0 .. n-Lp */
        for (int s = 0; s < S; ++s) nm[s] = INFINITY;
        /* branches in list order: start state ascending, then input symbol ascending; strict '<'
         * keeps the first listed candidate on a tie (algorithm.py:79-83) */
        for (int s = 0; s < S; ++s) {
            const int corr = s / NC;
            const int v = st->v[s];
            const int r = imod(2 * (int64_t)v - tilt, 2 * p);
            const double cr = rot_cs[2 * r], sr = rot_cs[2 * r + 1];
            for (int u = 0; u < M; ++u) {
                const int f = u + M * corr;
                const double inc = -fma(cr, Z[2 * f], sr * Z[2 * f + 1]);   /* -Re(e^{-j theta} Z) */
                const double cand = st->metric[s] + inc;
                const int u_old = Lp == 1 ? u : corr / msub;
                const int corr2 = Lp == 1 ? 0 : u + M * (corr % msub);
                const int v2 = (v + K_old * u_old) % p;
                const int s2 = v2 % NC + NC * corr2;
                if (cand < nm[s2]) {
                    nm[s2] = cand;
                    nv[s2] = v2;
                    nh_[s2] = (st->hist[s] << c->lgM) | (uint64_t)u;
                }
            }
        }
        double mn = nm[0];
        for (int s = 1; s < S; ++s) mn = nm[s] < mn ? nm[s] : mn;
        int best = -1;
        for (int s = 0; s < S; ++s) {
            st->metric[s] = nm[s] - mn;                         /* algorithm.py:65-67 */
            st->v[s] = nv[s];
            st->hist[s] = nh_[s];
            if (best < 0 && nm[s] == mn) best = s;              /* np.argmin: first minimum (:92) */
        }
        if (n >= D - 1) out_syms[n - D + 1] = (uint8_t)((st->hist[best] >> (c->lgM * (D - 1))) & (uint64_t)(M - 1));
        st->n = n + 1;
    }
    return 0;
}
