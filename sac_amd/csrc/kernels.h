// sac_amd/csrc/kernels.h -- launch wrappers of the HIP kernels (definitions in kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "params.h"

namespace sacamd {

struct PcmView {            // centred planar int32 PCM of the staged batch
  const int *pcm;
  long long frame_stride, ch_stride;
  unsigned long long *prof;   // optional: 8 section cycle counters of the OLS kernel (debug)
  double *keep;               // kept p_lpc streams of earlier search generations (WorkItem::pin_kept), nullable
  int *progress;              // final pass only (else null): per work-item count of p_lpc samples the OLS kernel has produced; the
                              // cascade kernel of the item runs at the same time and follows it chunk by chunk
  int *started;               // with progress: number of OLS workgroups that have begun (the host launches the cascade after all have)
  int hiprio;                 // 1: latency-bound launch (final pass): its waves raise their issue priority (s_setprio) so that
                              // throughput work sharing the CU (another batch's search) fills the gaps instead of slowing them
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
                 const double *d_p, int *d_err, int *d_pred /*nullable*/, int *d_nonfinite /*[count]: set to 1 where the prediction was not finite*/);
// ---- costs / s2u (kernels_misc.hip)
void launch_cost(hipStream_t s, int kind, const int *d_err, const long long *d_off, const int *d_n, int count,
                 int *d_hist_scratch, double *d_cost);
void launch_s2u(hipStream_t s, const int *d_err, int *d_s2u, const long long *d_off, const int *d_n, int count, int *d_maxbpn);
size_t cost_hist_scratch_ints();
// SparsePCM::Analyse sums per block: out4[4b..] = {sum|val|, sum|rank|, used values, range (-1: unsupported)}
void launch_sparse_cost(hipStream_t s, const int *d_pcm, const long long *d_off, const int *d_n, int count, long long *d_out4);
// ---- coder (kernels_coder.hip)
struct CoderJob {
  long long off_in;     // ints into d_s2u
  long long off_out;    // bytes into d_out
  int n, maxbpn;
  int cap;              // output capacity in bytes
  int with_map;         // 1: MapEncoder prefix over used flags
  long long off_used;   // bytes into d_used (usedl at +0 .. usedh at +32769)
};
void launch_coder(hipStream_t s, const CoderJob *d_jobs, int count, const int *d_s2u, const int *d_s2u_map /*jobs with_map*/, const unsigned char *d_used,
                  const unsigned short *d_laplace, const short *d_fwd, const unsigned short *d_inv,
                  unsigned char *d_state, size_t state_stride, unsigned char *d_out, int *d_len, int hiprio = 0);
size_t coder_state_bytes();
struct RemapJob {
  long long off;        // ints into pred / err / s2u_map planes
  long long off_used;   // bytes into d_used
  int n;
};
void launch_remap(hipStream_t s, const RemapJob *d_jobs, int count, const unsigned char *d_used, const int *d_pred, const int *d_err,
                  int *d_s2u_map, int *d_prefix_scratch, long long *d_out3);

}  // namespace sacamd
