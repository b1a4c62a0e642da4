// sac_amd/csrc/params.h -- per-(frame x candidate x predictor-slot) work-item description.
//
// Host side: the 58-coefficient profile -> predictor parameters mapping of
// FrameCoder::SetParam (/root/reference/src/libsac/libsac.cpp:37-92) and the regressor geometry
// of Predictor::fillbuf_ch0/ch1 (libsac/pred.cpp:17-31) + the stereo schedule of
// FrameCoder::PredictFrame (libsac.cpp:113-141), flattened into plain-old-data that the three
// predictor kernels consume.
#pragma once
#include <cmath>
#include <cstdint>

namespace sacamd {

constexpr int kNumCoefs = 58;
constexpr int kMaxOLS = 128;    // regressor length limit of the kernels (profile box: <= 96)
constexpr int kStages = 4;
// regressor-length capacity of each OLS kernel instance; the last one is the two-wave generic path
constexpr int kNumOlsClasses = 8;
constexpr int kOlsClassMax[kNumOlsClasses] = {16, 24, 32, 40, 48, 56, 64, 96};
constexpr int kNumLmsClasses = 16;   // 0..6 and (round 6) 14, 15: search layouts (free summation order); 7..9 canonical-order layouts (systolic, stage by stage), 10..13 lane-map canonical layouts (final pass)
constexpr int kLmsCanonFirst = 7;
constexpr int kLmsCanon3First = 10;   // lane-map canonical layouts (no lane-major table copies: off_tabc stays -1)

struct ChanParam {
  // OLS
  int n_ols;          // regressor length a+b+c
  int k;              // solve every k updates
  int a, b, c, du;    // x = [self[t-a..t-1], other[u-b..u+c-1]], u = max(0, t-du)
  double lambda, nu_eff, beta_sum, beta_pow, beta_add;
  // cascade
  int vn[kStages];
  double vmu[kStages], vmudecay[kStages], vpowdecay[kStages];
  double mu_mix, mu_mix_beta;
  int lm_n;
  double lm_alpha, proj_alpha;
  int lo, hi;         // Cascade clamp range (framestats of the predictor *slot*, libsac.cpp:99-102)
  // bias
  double bias_mu;
  int bias_scale;
  int out_lo, out_hi; // output clamp (framestats of the *file* channel, libsac.cpp:106)
};

struct WorkItem {
  int frame;          // frame slot in the context
  int ch_self;        // file channel this work-item predicts
  int ch_other;       // file channel of the "other" regressor part (== ch_self for mono)
  int slot;           // predictor slot 0/1
  int start, n;       // window [start, start+n) inside the frame
  int lms_class;      // register layout class of the cascade kernel (kernels_pred.hip, LmsCfg); >= kLmsCanonFirst: canonical summation order
  int ols_class;      // index into kOlsClassMax (LDS capacity class of the OLS kernel)
  long long off_p;    // doubles: this item's p_lpc stream in the OLS buffer and its p_lpc+p_lms stream in the cascade buffer [n]
  long long off_pin;  // doubles: where the cascade reads p_lpc (== off_p unless the OLS result is shared with another item)
  int pin_kept;       // 1: off_pin is relative to the context's buffer of kept search-window streams, not to the p_lpc buffer
  int ols_item;       // index of the work-item whose OLS stage produces this item's p_lpc (itself unless shared)
  long long off_err;  // int32 residual [n]
  long long off_tab;  // doubles: per stage {mutab[vn], powtab[vn]}, stages back to back
  long long off_tabc; // doubles: lane-major copies for the canonical-order layouts (pred_tables.h), -1 = none
  double sum_powtab[kStages];   // filled by the table kernel
  ChanParam p;
};

struct FrameStatsD {
  int mean, minval, maxval;   // min/max already shifted by mean
  int numsamples;
};

static inline int iround_f(float v) { return (int)std::round((double)v); }

// libsac.cpp:37-92 for one candidate; slot 0 and slot 1 parameter blocks.
// nch==1: only slot 0 is meaningful.  stats[ch] = framestats of file channel ch.
static inline void map_profile(const float *g, bool optimize, int optk, int nch,
                               const FrameStatsD *stats, ChanParam out[2], int *ch_ref_out) {
  ChanParam &p0 = out[0], &p1 = out[1];
  const int k = optimize ? optk : 1;
  p0.k = p1.k = k;
  p0.lambda = g[0]; double nu0 = g[1];
  p1.lambda = g[12]; double nu1 = g[13];
  p0.nu_eff = (1.0 - p0.lambda) * nu0;     // pred/ols.cpp:11
  p1.nu_eff = (1.0 - p1.lambda) * nu1;
  const int n0[4] = {28, 29, 30, 37}, n1[4] = {31, 32, 33, 38};
  const int mu0[4] = {2, 3, 4, 5}, mu1[4] = {14, 15, 16, 17};
  const int md0[4] = {6, 39, 46, 47}, md1[4] = {18, 40, 48, 49};
  const int pd0[4] = {7, 8, 50, 51}, pd1[4] = {19, 20, 21, 52};
  for (int i = 0; i < 4; i++) {
    p0.vn[i] = iround_f(g[n0[i]]); p1.vn[i] = iround_f(g[n1[i]]);
    p0.vmu[i] = (double)g[mu0[i]] / double(p0.vn[i]);
    p1.vmu[i] = (double)g[mu1[i]] / double(p1.vn[i]);
    p0.vmudecay[i] = g[md0[i]]; p1.vmudecay[i] = g[md1[i]];
    p0.vpowdecay[i] = g[pd0[i]]; p1.vpowdecay[i] = g[pd1[i]];
  }
  p0.mu_mix = g[10]; p0.mu_mix_beta = g[11];
  p1.mu_mix = g[22]; p1.mu_mix_beta = g[23];
  const int nA = iround_f(g[24]), nB = iround_f(g[25]), nS0 = iround_f(g[26]);
  int nS1 = iround_f(g[27]);
  const int nM0 = iround_f(g[9]);
  p0.beta_sum = g[34]; p0.beta_pow = g[35]; p0.beta_add = g[36];
  p1.beta_sum = g[53]; p1.beta_pow = g[54]; p1.beta_add = g[55];
  p0.proj_alpha = g[56]; p1.proj_alpha = g[57];
  p0.lm_n = p1.lm_n = iround_f(g[41]);
  p0.lm_alpha = p1.lm_alpha = g[42];
  p0.bias_mu = g[43]; p1.bias_mu = g[44];
  p0.bias_scale = p1.bias_scale = iround_f(g[45]);
  int ch_ref = 0;
  if (nS1 < 0) { nS1 = -nS1; ch_ref = 1; }
  *ch_ref_out = ch_ref;
  // regressor geometry
  p0.a = nA; p0.b = nM0; p0.c = 0;
  p0.du = (nch == 2) ? ((nS1 > 1 ? nS1 : 1) - 1) : 0;
  p1.a = nB; p1.b = nS0; p1.c = nS1; p1.du = 0;
  p0.n_ols = p0.a + p0.b + p0.c;
  p1.n_ols = p1.a + p1.b + p1.c;
  // ranges: cascade clamp by slot (bug-compatible), output clamp by file channel
  const int f0 = (nch == 2) ? ch_ref : 0, f1 = 1 - f0;
  p0.lo = stats[0].minval; p0.hi = stats[0].maxval;
  p0.out_lo = stats[f0].minval; p0.out_hi = stats[f0].maxval;
  if (nch == 2) {
    p1.lo = stats[1].minval; p1.hi = stats[1].maxval;
    p1.out_lo = stats[f1].minval; p1.out_hi = stats[f1].maxval;
  } else {
    p1.lo = p0.lo; p1.hi = p0.hi; p1.out_lo = p0.out_lo; p1.out_hi = p0.out_hi;
  }
}

}  // namespace sacamd
