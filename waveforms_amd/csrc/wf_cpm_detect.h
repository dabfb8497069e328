// wf_cpm_detect.h — what the two forms of the generic CPM trellis detector share (wf_cpm_detect.hip: one 16-lane DPP
// row per chunk, any trellis of up to 16 states; wf_cpm_lanes.hip: one LANE per chunk, trellis known at compile time).
// Both write the same per-chunk proof records, so one verify kernel and one repair kernel serve either.
#pragma once
#include "wf_common.h"

// Device-resident detector state (WF_CPM_STATE_BYTES), trellises of up to 16 states: words 0..63 current, 64..127 staging
// (17 .. 64 states: the wide layout of wf_cpm_wide.hip in the same block).
//   [0] calls made (as int64), [1..16] metrics (double), [17..32] tilted phase indices r (int64),
//   [33..48] decision registers (uint64)
#define CPM_ST_N 0
#define CPM_ST_M 1
#define CPM_ST_V 17
#define CPM_ST_H 33
#define CPM_ST_STAGE 64

// proof record per CHUNK: [0] the state its own calls started from, [1] the state it ended with; 16 lanes x 3
// words each (metric, phase index, decision register).  cpm_verify_kernel compares chunk c's start with chunk
// c - 1's end.
#define CPM_EDGE_WORDS (2 * 16 * 3)
// Repair lists, behind the proof records of a launch (all three forms): CPM_NLIST counts, then CPM_NLIST lists of
// nchunks entries (a chunk index each; a list can never overflow: a round lists a chunk at most once).
//   list 0: chunks whose start differs from their predecessor's end (the verify kernel);
//   list 1, 2: chunks a repair of round 1 / 2 handed on to (the chunk behind one whose END changed);
//   lists 2 <-> 3: the finisher's rounds — one workgroup that repeats until a round hands nothing on.
// Invariant kept by every repair: a chunk's record (start, end) and its decisions are those of ONE run of its calls
// from `start`.  A repair reads the predecessor's current end, makes it the chunk's start, runs the calls again
// (beside the previous run from the old start, until the two are bitwise equal: from there on nothing changes) and,
// if the chunk's end changed, lists the next chunk.  Each round's smallest listed chunk is repaired from the true
// state, so the consistent prefix grows every round: worst case the rounds are the sequential detector.
#define CPM_NLIST 4
#define CPM_REPAIR_BLOCKS 256   // workgroups of a parallel repair round (grid-stride over the list)
#define CPM_ROT_SIN 128         // rotation table in LDS: cos at [r], sin at [CPM_ROT_SIN + r], r < 2p <= 128

__host__ __device__ inline uint64_t *cpm_list_counts(uint64_t *edge, int64_t nchunks, int edge_words) { return edge + nchunks * edge_words; }
__host__ __device__ inline uint64_t *cpm_list(uint64_t *edge, int64_t nchunks, int edge_words, int k)
{
    return edge + nchunks * edge_words + CPM_NLIST + (int64_t)k * nchunks;
}
__host__ __device__ inline size_t cpm_edge_total_words(int64_t nchunks, int edge_words)
{
    return (size_t)nchunks * edge_words + CPM_NLIST + (size_t)CPM_NLIST * nchunks;
}

// (M - 1) * sum of K over symbols 0 .. n - Lp, mod 2p: the phase tilt of call n
__host__ __device__ inline int cpm_tilt(int M, int p, int nh, int K0, int K1, int Lp, int64_t n)
{
    const int64_t m = n - Lp + 1;
    if (m <= 0) return 0;
    const int per = nh == 2 ? K0 + K1 : K0;
    int64_t acc = (m / nh) % (2 * p) * per;
    if (nh == 2 && (m & 1)) acc += K0;
    return (int)(((int64_t)(M - 1) * (acc % (2 * p))) % (2 * p));
}

// Lane-per-chunk form (wf_cpm_lanes.hip).  wf_cpm_lanes_plan: 0 and the chunk length / resident-wave geometry when a
// compiled specialisation exists for this detector, 1 when not (the caller runs the generic kernel).
struct cpm_lane_plan {
    int spec;           // which specialisation
    int ring_batches;   // LDS ring depth in 16 KB batches
    int waves_per_cu;   // resident waves per CU the ring allows
    int calls_per_batch;
    int min_chunk;      // shortest chunk worth running (calls): below it the warm-up's share of the work and its re-read of the rows outweigh the extra waves
    double lane_ns_per_call;    // the lane form's time per call of a chunk (a lane runs its chunk alone: independent of the burst)
    double row_ns_per_call;     // the row form's time per call of the BURST (measured at 1e7 calls; it scales with the burst)
};
int wf_cpm_lanes_plan(const wf_cpm_detector_config *det, cpm_lane_plan *plan);   // 0: a specialisation exists (*plan filled in), 1: none
// The matched filters INSIDE the detector (round 6): d_rows_ri of the launch are then the noisy samples, call k's window = samples
// 8 k .. 8 k + 8 from that pointer (nsamp addressable), against template column (k + col0) % nh of d_templates [nh][16][9] complex,
// whose filters f and 15 - f are exact conjugates (the caller checked).  Lane form of the 16-filter ARTM design; the repairs
// (cpm_repair_kernel) rebuild the rows of the chunks they run from the same samples with the same arithmetic.
struct cpm_mf_source {
    const double *d_templates;
    int64_t nsamp;
    int col0;
    int64_t start0 = 0;     // quad form only (the lane form is handed the pointer to its first window): window of call k starts at sample start0 + 8 k; samples outside [0, nsamp) count as zero
};
int wf_cpm_lanes_launch(const cpm_lane_plan &plan, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri,
                        int64_t ncalls, int warmup, int chunk_calls, int64_t nchunks, uint8_t *d_decisions, void *d_state, uint64_t *d_edge,
                        void *stream, int64_t slack_lo_bytes, int64_t slack_hi_bytes, bool solo, const cpm_mf_source *mf = nullptr);
// wf_cpm_viterbi_detect for callers that own the memory around the rows (the links: rows sit inside their workspace):
// slack_*_bytes of it before / behind the array may be READ (never interpreted) by the lane form's row fetch.
int wf_cpm_viterbi_detect_in(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri, int64_t ncalls,
                             int warmup, uint8_t *d_decisions, void *d_state, void *stream, int64_t slack_lo_bytes, int64_t slack_hi_bytes,
                             bool beside = false,    // beside: the launch will share the chip with another kernel (a pipelined link's front end)
                             const cpm_mf_source *mf = nullptr,    // mf: d_rows_ri are the noisy samples, the matched filters run inside the detector
                             int edge_slot = -1);   // 0 / 1: which of two sets of proof records in the context this launch uses (launches in flight on two streams); -1: the one set
int wf_cpm_samples_form_chunk(wf_ctx *ctx, const wf_cpm_detector_config *det, int64_t ncalls, int warmup, bool beside);
bool wf_cpm_samples_form_applies(wf_ctx *ctx, const wf_cpm_detector_config *det, int64_t ncalls, int warmup, int sps, int nfilt, int ntm, int64_t start0);

// Wide form (wf_cpm_wide.hip): trellises of 17 .. 64 states, lane = state, one wave = one detector.  Proof records of
// 2 x 64 x 3 words per chunk, detector state in the wide layout of WF_CPM_STATE_BYTES.
int wf_cpm_wide_applies(const wf_cpm_detector_config *det);
int64_t wf_cpm_wide_chunk_calls(int64_t ncalls, int W, int cus, int64_t chunk_opt);
int wf_cpm_wide_warmup(int warmup);
int wf_cpm_wide_detect(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri, int64_t ncalls,
                       int warmup, uint8_t *d_decisions, void *d_state, void *stream);

// Quad form (wf_cpm_quad.hip): trellises of 65 .. 256 states (pulse of 2 or 3 symbols), thread = state, one workgroup of four
// waves = one detector.  Proof records of 2 x 256 x 3 words per chunk, detector state in the quad layout of WF_CPM_STATE_BYTES.
int wf_cpm_quad_applies(const wf_cpm_detector_config *det);
int64_t wf_cpm_quad_chunk_calls(int64_t ncalls, int W, int cus, int64_t chunk_opt);
int wf_cpm_quad_warmup(int warmup);
int wf_cpm_quad_detect(wf_ctx *ctx, const wf_cpm_detector_config *det, const double *d_rot_cs, const double *d_rows_ri, int64_t ncalls,
                       int warmup, uint8_t *d_decisions, void *d_state, void *stream, const cpm_mf_source *mf = nullptr);   // mf: d_rows_ri are the noisy samples (64 filters of 9 taps formed per batch by the detector's own threads)
