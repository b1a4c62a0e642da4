// sac_amd/csrc/pred_bias.h -- stage 3 of the predictor: bias correction + residual.
//
// Reference: BiasEstimator (/root/reference/src/pred/bias.h:16-175) with its SSLMS mixers
// (pred/ls.h:279-292) and RunMeanVar (common/utils.h:75-110), then the rounding/clamping of
// FrameCoder::PredictFrame's eprocess lambda (libsac/libsac.cpp:104-110).
//
// Input is the p_lpc+p_lms stream of stage 2 plus the PCM; it is a ~150-operation scalar
// recurrence per sample, so one LANE runs one work-item (64 work-items per wave); the three
// context tables (32+8+16 entries of {cnt,val}) sit in LDS, one private slab per lane.
#pragma once
#include "canon.h"
#include "libm_port.h"
#include "params.h"

namespace sacamd {

constexpr int kBiasCtx = 56;                       // 32 (ctx0) + 8 (ctx1) + 16 (ctx2)
constexpr int kBiasSlabDoubles = 2 * kBiasCtx + 12;   // {cnt,val} pairs + 4x3 SSLMS mixer weights

// tables: this lane's slab of kBiasSlabDoubles doubles.  err/pred may be null.  *nonfinite (nullable) is set to 1
// when a prediction is not finite -- where the reference throws (pred/cascade.h:40-41) the caller reports
// SACAMD_ERR_NONFINITE (final pass) or an infinite cost (search).
// Remap::Unmap (map.cpp:189-202) with prefix counts instead of the value-by-value walk: the smallest step err >= 1 in the
// direction of merr's sign such that exactly |merr| used values lie in (pred, pred + err]; 0 and everything the walk would
// never leave (no such step) -> the walk's bound.  prefix[j] = #used values in [-32768, -32768 + j - 1], j = 0 .. 65537.
SA_HD int bias_unmap(const int *prefix, int pred, int merr) {
  if (merr == 0) return 0;
  const int sgn = merr < 0 ? -1 : 1, m = merr < 0 ? -merr : merr;
  auto cnt = [&](int a, int b) {          // used values in [a, b]
    a = a < -32768 ? -32768 : a; b = b > 32768 ? 32768 : b;
    return a > b ? 0 : prefix[b + 32768 + 1] - prefix[a + 32768];
  };
  int lo = 1, hi = 1 << 17;               // smallest e with count(e) >= m (the count grows by at most one per step)
  while (lo < hi) {
    const int e = (lo + hi) >> 1;
    const int c = sgn > 0 ? cnt(pred + 1, pred + e) : cnt(pred - e, pred - 1);
    if (c >= m) hi = e; else lo = e + 1;
  }
  return sgn * lo;
}

// dec != null: decoder (FrameCoder::UnpredictFrame's dprocess, libsac.cpp:153-165): the sample is the prediction plus the
// decoded residual (un-mapped first for mapped streams), written to dec->self_w and announced; `self` is not read.
SA_HD void bias_stage(const ChanParam &p, const int *self, int n, const double *psum, int mean,
                      int *err, int *pred, double *tables, int *nonfinite = nullptr, const DecLink *dec = nullptr, double *pd_out = nullptr) {
  double *cnt = tables, *val = tables + kBiasCtx;
  for (int i = 0; i < kBiasCtx; i++) { cnt[i] = 4.0; val[i] = 0.0; }
  double *mixw = tables + 2 * kBiasCtx;    // [4][3]
  for (int a = 0; a < 12; a++) mixw[a] = 0.0;
  double hin0 = 0, hin1 = 0, hin2 = 0;
  double hd0 = 0, hd1 = 0, hd2 = 0, hd3 = 0, hd4 = 0;
  double rmean = 0.0, rvar = 0.0;
  const double nscale = (double)(1 << p.bias_scale);
  const double mu = p.bias_mu;

  for (int t = 0; t < n; t++) {
    if (dec && !sa_wait_ge(dec->prog_in, t + 1, dec->fail)) return;
    const double px = psum[t];
    if (nonfinite && !(fabs(px) <= 1.79769313486231570815e308)) *nonfinite = 1;
    // CalcContext (bias.h:64-113)
    const int b0 = hin0 > px ? 0 : 1;
    const int b2 = hd0 < 0 ? 0 : 1, b3 = hd1 < 0 ? 0 : 1, b4 = hd2 < 0 ? 0 : 1;
    const int b5 = hd1 < hd0 ? 0 : 1, b6 = hd2 < hd1 ? 0 : 1, b7 = hd3 < hd2 ? 0 : 1, b8 = hd4 < hd3 ? 0 : 1;
    const int b9 = fabs(hd0) > 32 ? 0 : 1;
    const int b10 = 2 * hin0 - hin1 > px ? 0 : 1;
    const int b11 = 3 * hin0 - 3 * hin1 + hin2 > px ? 0 : 1;
    double sum = 0;
    sum += fabs(hd0); sum += fabs(hd1); sum += fabs(hd2); sum += fabs(hd3); sum += fabs(hd4);
    sum /= 5.0;
    int mix_ctx = 0;
    if (sum > 512) mix_ctx = 2; else if (sum > 32) mix_ctx = 1;
    const int c0 = b0 + (b2 << 1) + (b9 << 2) + (b10 << 3) + (b11 << 4);
    const int c1 = 32 + b2 + (b3 << 1) + (b4 << 2);
    const int c2 = 40 + b5 + (b6 << 1) + (b7 << 2) + (b8 << 3);
    // Predict (bias.h:114-126)
    double pt[3];
    pt[0] = val[c0] / cnt[c0];
    pt[1] = val[c1] / cnt[c1];
    pt[2] = val[c2] / cnt[c2];
    double *mw = mixw + 3 * mix_ctx;
    const double pbias = dot_canon(pt, mw, 3);
    const double pd = px + pbias;
    if (pd_out) pd_out[t] = pd;                      // Predictor::predict's return value (libsac/pred.cpp:33-38), before eprocess rounds it
    // eprocess (libsac.cpp:105-109)
    const int pi = clampi32(cvt_i32_x86(round(pd)), p.out_lo, p.out_hi);
    int v;
    if (dec) {
      const int e = dec->merr[t];
      v = pi + (dec->prefix ? bias_unmap(dec->prefix, pi + mean, e) : e);
      dec->self_w[t] = v;
      sa_publish(dec->prog_out, t + 1);
    } else v = self[t];
    if (pred) pred[t] = pi + mean;
    if (err) err[t] = v - pi;
    // Update (bias.h:127-163)
    const double dv = (double)v;
    const double delta = dv - round(px);
    hin2 = hin1; hin1 = hin0; hin0 = dv;
    hd4 = hd3; hd3 = hd2; hd2 = hd1; hd1 = hd0; hd0 = delta;
    const double var0 = fmax(0.0, rvar);
    const double diff = delta - rmean;
    const double z = diff * diff / (var0 + 1E-5);
    const double w = sa_exp(-0.5 * z);
    const int cc[3] = {c0, c1, c2};
    for (int q = 0; q < 3; q++) {
      const int c = cc[q];
      double vv = val[c] + w * delta;      // not fused in the reference binary
      double cn = cnt[c] + w;
      if (cn >= nscale) { vv *= 0.5; cn *= 0.5; }
      val[c] = vv; cnt[c] = cn;
    }
    const double a = 0.998;
    const double old_mean = rmean;
    rmean = fma(a, rmean, (1.0 - a) * delta);
    rvar = fma(a, rvar, (1.0 - a) * ((delta - old_mean) * (delta - rmean)));
    const double wf = mu * sgnd(delta - pbias);
    for (int i = 0; i < 3; i++) mw[i] = fma(wf, sgnd(pt[i]), mw[i]);
  }
}

}  // namespace sacamd
