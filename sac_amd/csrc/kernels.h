// sac_amd/csrc/kernels.h -- launch wrappers of the HIP kernels (definitions in kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "params.h"
#include "simt.h"

namespace sacamd {

struct PcmView {            // centred planar int32 PCM of the staged batch
  const int *pcm;
  long long frame_stride, ch_stride;
  unsigned long long *prof;   // optional: 8 section cycle counters of the OLS kernel (debug)
  double *keep;               // kept p_lpc streams of earlier search generations (WorkItem::pin_kept), nullable
};

// ---- analyse (kernels_misc.hip)
void launch_analyse_s16(hipStream_t s, int nframes, int nch, const int16_t *d_il, const long long *d_frame_off,
                        const int *d_nsamp, int zero_mean, int *d_pcm, long long frame_stride, long long ch_stride,
                        FrameStatsD *d_stats, unsigned char *d_used /*nullable [f][ch][65537]*/);
void launch_analyse_i32(hipStream_t s, int nframes, int nch, const int *d_raw, long long raw_fs, long long raw_cs,
                        const int *d_nsamp, int zero_mean, int *d_pcm, long long frame_stride, long long ch_stride,
                        FrameStatsD *d_stats, unsigned char *d_used);
// ---- predictor stages (kernels_pred.hip)
void launch_tables(hipStream_t s, WorkItem *d_items, int count, double *d_tab);
void launch_ols(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, int ols_class, PcmView v, double *d_p, bool latency_bound);
struct LmsRingCap { int c[4]; };   // per-stage history ring capacity (doubles) of one launch
size_t lms_lds_bytes(int lms_class, const LmsRingCap &rc);
int lms_class_for(const int *vn, bool canon);   // cascade layout class for stage lengths vn[4]; canon: slmath::dot summation order (final pass)
int lms_max_wg_per_cu(int lms_class);          // register-file bound on resident workgroups per CU
void launch_lms(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, int lms_class, LmsRingCap rc, PcmView v,
                const double *d_tab, const double *d_p /*p_lpc in*/, double *d_q /*p_lpc+p_lms out*/);
void launch_bias(hipStream_t s, const WorkItem *d_items, int count, PcmView v, const FrameStatsD *d_stats, int nch,
                 const double *d_p, int *d_err, int *d_pred /*nullable*/, int *d_nonfinite /*[count]: set to 1 where the prediction was not finite*/,
                 double *d_pd = nullptr /*nullable: the bias stage's prediction before rounding (Predictor::predict), laid out like d_p*/);
// ---- decoder: the three stages of every channel running side by side, two launches per group of frames (kernels_pred.hip)
int dec_lms_class_for(const int *vn);            // cascade layout class of a decoder work-item (256-lane layouts only)
size_t dec_cascade_lds_bytes(int lms_class, const LmsRingCap &rc);
void launch_dec_cascade(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, size_t lds_bytes, LmsRingCap rc, PcmView v,
                        const double *d_tab, const double *d_p, double *d_q, const DecLink *d_links, int *d_started /*host-visible*/);
void launch_dec_olsbias(hipStream_t s, const WorkItem *d_items, const int *d_idx, int n_ols, int n_bias, bool any_wide, PcmView v, double *d_p, const double *d_q,
                        const FrameStatsD *d_stats, int nch, const DecLink *d_lk_ols, const DecLink *d_lk_bias);
// the same group as ONE launch (3 m blocks, one CU each): co-residency by construction, no concurrent queues needed
void launch_dec_all(hipStream_t s, const WorkItem *d_items, const int *d_idx, int m, size_t lds_cascade, bool any_wide, LmsRingCap rc, PcmView v,
                    const double *d_tab, double *d_p, double *d_q, const FrameStatsD *d_stats, int nch, const DecLink *d_lk_lms, const DecLink *d_lk_ols,
                    const DecLink *d_lk_bias);
void launch_used_prefix(hipStream_t s, int count, const unsigned char *d_used, const long long *d_off_used, int *d_prefix);
// ---- costs / s2u (kernels_misc.hip)
void launch_cost(hipStream_t s, int kind, const int *d_err, const long long *d_off, const int *d_n, int count,
                 int *d_hist_scratch, double *d_cost, const long long *d_hist_off = nullptr /*per-vector histogram start (ints)*/,
                 const long long *d_hist_cap = nullptr /*and capacity; default: count x cost_hist_scratch_ints()*/);
void launch_s2u(hipStream_t s, const int *d_err, int *d_s2u, const long long *d_off, const int *d_n, int count, int *d_maxbpn);
size_t cost_hist_scratch_ints();
// SparsePCM::Analyse sums per block: out4[4b..] = {sum|val|, sum|rank|, used values, range}; range -1: the block's value range
// (then in out4[4b+2]) exceeds the LDS bitmap -> run those blocks again with d_wide (per block: start and size, in 32-bit
// words, of its bitmap + prefix scratch in d_scratch; needs 2 * ceil(range / 32) + 1 words)
void launch_sparse_cost(hipStream_t s, const int *d_pcm, const long long *d_off, const int *d_n, int count, long long *d_out4,
                        unsigned *d_scratch = nullptr, const long long *d_wide = nullptr);
// ---- coder (kernels_coder.hip)
struct CoderJob {
  long long off_in;     // ints into d_s2u
  long long off_out;    // bytes into d_out
  int n, maxbpn;        // maxbpn < 0: taken from the device array d_maxbpn[-1 - maxbpn] (written by k_s2u earlier on the stream)
  int cap;              // output capacity in bytes
  int with_map;         // 1: MapEncoder prefix over used flags
  long long off_used;   // bytes into d_used (usedl at +0 .. usedh at +32769)
};
void launch_coder(hipStream_t s, const CoderJob *d_jobs, int count, const int *d_s2u, const int *d_s2u_map /*jobs with_map*/, const unsigned char *d_used,
                  const unsigned short *d_laplace, const short *d_fwd, const unsigned short *d_inv,
                  unsigned char *d_state, size_t state_stride, unsigned char *d_out, int *d_len,
                  const int *d_maxbpn = nullptr /*for jobs with maxbpn < 0*/,
                  unsigned char *d_compact = nullptr /*nullable: every finished stream appends its payload here (16-byte aligned)*/,
                  long long *d_compact_at = nullptr /*[count] offset of each payload in d_compact, [count] = bytes used (zeroed by the launcher)*/);
size_t coder_state_bytes();
void launch_libm_tap(hipStream_t s, int kind, const double *d_x, const double *d_y, int n, double *d_out);   // parity tap (sacamd_debug_libm)
struct DecJob {
  long long off_in;     // bytes into d_in (the channel's payload)
  long long off_out;    // ints into d_err
  int inlen, n, maxbpn;
  int with_map;         // 1: the payload starts with the MapEncoder header; the decoded flags go to d_used + off_used
  long long off_used;
};
void launch_decoder(hipStream_t s, const DecJob *d_jobs, int count, const unsigned char *d_in, int *d_err /*signed (mapped) residuals*/, unsigned char *d_used,
                    const unsigned short *d_laplace, const short *d_fwd, const unsigned short *d_inv,
                    unsigned char *d_state, size_t state_stride, int *d_consumed);
struct RemapJob {
  long long off;        // ints into pred / err / s2u_map planes
  long long off_used;   // bytes into d_used
  int n;
};
void launch_remap(hipStream_t s, const RemapJob *d_jobs, int count, const unsigned char *d_used, const int *d_pred, const int *d_err,
                  int *d_s2u_map, int *d_prefix_scratch, long long *d_out3);

}  // namespace sacamd
