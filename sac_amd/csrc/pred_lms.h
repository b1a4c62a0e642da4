// sac_amd/csrc/pred_lms.h -- stage 2 of the predictor: the Cascade.
//
// Reference: Cascade (/root/reference/src/pred/cascade.h:75-131) = 4 x NLMS_Stream
// (pred/ls.h:31-62) + RLS/ALC (pred/rls.{h,cpp}) blended by BlendLS (cascade.h:11-57) over two
// LS_ADA experts (ls.h:214-241) and BlendExp<RunSumEMA> (blend.h).  Its target is the OLS
// residual val - p_lpc (libsac/pred.cpp:43), so given the p_lpc stream of stage 1 it is again a
// self-contained recurrence.
//
// One workgroup of NL (=256) lanes per work-item.  Tap state (w, mutab, powtab) is register
// resident: tap i of stage s lives in lane i % NL, slot i / NL.  The four histories are rings
// in LDS.  Per sample ONE fused sweep does "weight update of step t-1" and "dot + power sum of
// step t" (the update needs the pre-push history, which is the post-push history shifted by
// one), then a wave-shuffle / LDS reduction, then the serial mixer chain on wave 0.
//
// Arithmetic: the per-tap element operations are the reference's (explicit fma); the N-term
// dot / power sums are reduced in lane-then-tree order, NOT slmath::dot's AVX2 order -- a
// ~1e-14 relative perturbation of p_lms (tolerance 1e-9 in the tests).  The
// small dots of the serial chain use the canonical order (canon.h).
#pragma once
#include "canon.h"
#include "libm_port.h"
#include "params.h"

namespace sacamd {

template <int C0, int C1, int C2, int C3>
struct LmsClass {
  static constexpr int c0 = C0, c1 = C1, c2 = C2, c3 = C3;
  static constexpr int total = C0 + C1 + C2 + C3;
  SA_HD static constexpr int slots(int s) { return s == 0 ? C0 : s == 1 ? C1 : s == 2 ? C2 : C3; }
  SA_HD static constexpr int first(int s) { return s == 0 ? 0 : s == 1 ? C0 : s == 2 ? C0 + C1 : C0 + C1 + C2; }
};
constexpr int kLmsChunk = 256;   // samples staged per global<->LDS exchange (== NL)
constexpr int kRlsMax = 10;

template <int N> struct DArr { double v[N]; };

// ---- serial mixer chain state (kept in LDS; wave 0 only, uniform)
struct LmsChain {
  double exw[2][5], exeg[2][5];   // LS_ADA experts: weights, squared-gradient EMAs
  double smw[2], smrs[2];         // BlendExp weights and running scores
  double S0, S1;                  // ALC
  double p[5], ep[2], pred;
};
constexpr int kLmsChainDoubles = (int)(sizeof(LmsChain) / sizeof(double));

template <int NL, class C>
struct LmsLds {
  double *ring[4];
  double *part;     // [2][NL/64][8]
  double *bc;       // [8]: wgrad[4], unused
  double *pin, *pout;
  double *P, *rx, *rw, *rph;
  double *chain;    // LmsChain (serial mixer state), only wave 0 touches it
  int *sv;
  SA_HD static size_t bytes() {
    size_t d = 0;
    for (int s = 0; s < 4; s++) d += (size_t)C::slots(s) * NL + 1;
    d += 2 * (NL / 64) * 8 + 8 + 2 * kLmsChunk + kRlsMax * kRlsMax + 3 * kRlsMax + kLmsChainDoubles;
    return d * sizeof(double) + kLmsChunk * sizeof(int) + 16;
  }
  SA_HD void carve(char *base) {
    double *d = reinterpret_cast<double *>(base);
    for (int s = 0; s < 4; s++) { ring[s] = d; d += (size_t)C::slots(s) * NL + 1; }
    part = d; d += 2 * (NL / 64) * 8;
    bc = d; d += 8;
    pin = d; d += kLmsChunk; pout = d; d += kLmsChunk;
    P = d; d += kRlsMax * kRlsMax; rx = d; d += kRlsMax; rw = d; d += kRlsMax; rph = d; d += kRlsMax;
    chain = d; d += kLmsChainDoubles;
    sv = reinterpret_cast<int *>(d);
  }
};


template <class E, class C>
SA_HD void lms_stage(E &ex, const ChanParam &p, const double *sum_powtab, const double *tab,
                     const int *self, int n, double *pio, char *lds_base) {
  constexpr int NL = E::nl;
  constexpr int NW = NL / 64;
  static_assert(kLmsChunk == NL, "chunk staging assumes one element per lane");
  LmsLds<NL, C> L;
  L.carve(lds_base);

  typename E::template Reg<DArr<C::total>> W, MT, PT;
  typename E::template Reg<DArr<8>> acc;

  int ns[4], cap[4], pos[4];
  for (int s = 0; s < 4; s++) { ns[s] = p.vn[s]; cap[s] = ns[s] + 1; pos[s] = 0; }
  const int m = p.lm_n;

  // ---- init: tables -> registers, zero rings / weights
  ex.par([&](int l) {
    const double *tp = tab;
    for (int s = 0; s < 4; s++) {
      const int f = C::first(s);
      for (int j = 0; j < C::slots(s); j++) {
        const int tap = j * NL + l;
        const bool on = tap < ns[s];
        W[l].v[f + j] = 0.0;
        MT[l].v[f + j] = on ? tp[tap] : 0.0;
        PT[l].v[f + j] = on ? tp[ns[s] + tap] : 0.0;
      }
      tp += 2 * ns[s];
      for (int i = l; i < cap[s]; i += NL) L.ring[s][i] = 0.0;
    }
    if (l < 8) L.bc[l] = 0.0;
    for (int i = l; i < kRlsMax * kRlsMax; i += NL) L.P[i] = 0.0;
    if (l < kRlsMax) { L.rx[l] = 0.0; L.rw[l] = 0.0; L.rph[l] = 0.0; }
  });
  ex.sync();
  LmsChain &ch = *reinterpret_cast<LmsChain *>(L.chain);
  ex.leader([&]() {
    for (int i = 0; i < m; i++) L.P[i * m + i] = 1.0;
    for (int e = 0; e < 2; e++) {
      for (int i = 0; i < 5; i++) { ch.exw[e][i] = 1.0 / 5; ch.exeg[e][i] = 0.0; }
      ch.smw[e] = 0.5; ch.smrs[e] = 0.0; ch.ep[e] = 0.0;
    }
    ch.S0 = ch.S1 = 0.0; ch.pred = 0.0;
    for (int i = 0; i < 5; i++) ch.p[i] = 0.0;
  });
  ex.sync();

  const double lo = (double)p.lo, hi = (double)p.hi;

  for (int t0 = 0; t0 < n; t0 += kLmsChunk) {
    // ---- stage a chunk of p_lpc / target in, flush the previous chunk of p_lpc+p_lms out
    ex.par([&](int l) {
      if (t0 > 0) pio[t0 - kLmsChunk + l] = L.pout[l];
      if (t0 + l < n) { L.pin[l] = pio[t0 + l]; L.sv[l] = self[t0 + l]; }
    });
    ex.sync();
    const int tend = (n - t0 < kLmsChunk) ? n - t0 : kLmsChunk;
    for (int tt = 0; tt < tend; tt++) {
      const int par = tt & 1;
      // ---- A: fused sweep (update of previous step, predict of this step)
      ex.par([&](int l) {
        for (int s = 0; s < 4; s++) {
          const int f = C::first(s);
          const double wg = L.bc[s];
          double d = 0.0, sp = 0.0;
          const double *ring = L.ring[s];
          for (int j = 0; j < C::slots(s); j++) {
            const int tap = j * NL + l;
            if (j * NL < ns[s] && tap < ns[s]) {
              int in = pos[s] + tap; if (in >= cap[s]) in -= cap[s];
              int io = in + 1; if (io >= cap[s]) io -= cap[s];
              const double xo = ring[io], xn = ring[in];
              double w = fma(MT[l].v[f + j], wg * xo, W[l].v[f + j]);
              w = clampd(w, -10.0, 10.0);
              W[l].v[f + j] = w;
              d = fma(xn, w, d);
              sp = fma(PT[l].v[f + j], xn * xn, sp);
            }
          }
          acc[l].v[s] = d; acc[l].v[4 + s] = sp;
        }
      });
      ex.template wave_sum<8>(acc);
      ex.par([&](int l) {
        if ((l & 63) == 0) for (int q = 0; q < 8; q++) L.part[(par * NW + (l >> 6)) * 8 + q] = acc[l].v[q];
      });
      ex.sync();
      // ---- B: serial chain on wave 0
      ex.leader([&]() {
        double dots[4], spow[4];
        for (int q = 0; q < 4; q++) {
          double a = L.part[(par * NW) * 8 + q], b = L.part[(par * NW) * 8 + 4 + q];
          for (int w = 1; w < NW; w++) { a = a + L.part[(par * NW + w) * 8 + q]; b = b + L.part[(par * NW + w) * 8 + 4 + q]; }
          dots[q] = a; spow[q] = b;
        }
        const double plpc = L.pin[tt];
        const double target = (double)L.sv[tt] - plpc;
        // Cascade::Predict
        for (int i = 0; i < 4; i++) ch.p[i] = dots[i];
        const double rpx = dot_canon(L.rx, L.rw, m);
        ch.p[4] = rpx;
        for (int e = 0; e < 2; e++) ch.ep[e] = dot_canon(ch.p, ch.exw[e], 5);
        ch.pred = dot_canon(ch.ep, ch.smw, 2);
        L.pout[tt] = plpc + ch.pred;
        // Cascade::Update(target)
        double bp[5];
        double p_prefix = 0.0;
        for (int i = 0; i <= 4; i++) {
          const double ew[2] = {ch.exw[0][i], ch.exw[1][i]};
          const double wgt = fmax(dot_canon(ew, ch.smw, 2), 0.0);
          const double px = fma(1.0 - p.proj_alpha, p_prefix, p.proj_alpha * ch.pred);
          bp[i] = target - clampd(px, lo, hi);
          p_prefix = fma(wgt, ch.p[i], p_prefix);
        }
        for (int s = 0; s < 4; s++) {
          // NLMS_Stream::Update scalar part (ls.h:47-48)
          L.bc[s] = p.vmu[s] * (bp[s] - dots[s]) * sum_powtab[s] / (spow[s] + 1.0);
          int np = pos[s] - 1; if (np < 0) np += cap[s];
          L.ring[s][np] = bp[s];
        }
        // RLS::UpdateHist(bp[4]) (rls.cpp:28-65)
        {
          const double val = bp[4];
          const double err = val - rpx;
          for (int i = 0; i < m; i++) L.rph[i] = dot_canon(&L.P[i * m], L.rx, m);
          const double phi = fmax(dot_canon(L.rx, L.rph, m), 1e-8);
          const double err2 = err * err;
          const double R = fmax(ch.S0 - ch.S1, 1e-5);
          const double nis = err2 / (phi + R);
          const double mm = sa_exp(-p.lm_alpha * nis);
          const double alpha = fma(0.999 - 0.99, mm, 0.99);
          const double denom = 1. / (alpha + phi);
          const double inv_alpha = 1.0 / alpha;
          for (int i = 0; i < m; i++)
            for (int j = 0; j <= i; j++) {
              const double pm = L.rph[i] * L.rph[j];
              const double v = fma(-denom, pm, L.P[i * m + j]) * inv_alpha;
              L.P[i * m + j] = v; L.P[j * m + i] = v;
            }
          for (int i = 0; i < m; i++) L.rw[i] = fma(err, denom * L.rph[i], L.rw[i]);
          ch.S0 = fma(0.95, ch.S0, (1.0 - 0.95) * err2);
          ch.S1 = fma(0.95, ch.S1, (1.0 - 0.95) * phi);
          for (int i = m - 1; i > 0; i--) L.rx[i] = L.rx[i - 1];
          if (m > 0) L.rx[0] = val;
        }
        // BlendLS::Update: experts (L1 then L2), then BlendExp
        for (int e = 0; e < 2; e++) {
          const double error = target - ch.ep[e];
          const double loss = (e == 0) ? sgnd(error) : error;
          const double beta = p.mu_mix_beta, beta1 = 1.0 - p.mu_mix_beta;
          for (int i = 0; i < 5; i++) {
            const double grad = loss * ch.p[i];
            ch.exeg[e][i] = fma(beta, ch.exeg[e][i], beta1 * grad * grad);
            const double mu_scaled = p.mu_mix / (sqrt(ch.exeg[e][i]) + 1e-5);
            ch.exw[e][i] = fma(mu_scaled, grad, ch.exw[e][i]);
          }
        }
        {
          double zm[2];
          for (int e = 0; e < 2; e++) {
            const double loss = fabs(target - ch.ep[e]);
            ch.smrs[e] = fma(0.95, ch.smrs[e], (1.0 - 0.95) * (-loss));
            zm[e] = 1.0 * ch.smrs[e];
          }
          const double maxz = fmax(zm[0], zm[1]);
          const double w0 = sa_exp(zm[0] - maxz), w1 = sa_exp(zm[1] - maxz);
          const double inv = 1.0 / (w0 + w1);
          ch.smw[0] = w0 * inv; ch.smw[1] = w1 * inv;
        }
      });
      for (int s = 0; s < 4; s++) { pos[s] -= 1; if (pos[s] < 0) pos[s] += cap[s]; }
      ex.sync();
    }
  }
  // flush the last chunk
  ex.par([&](int l) {
    const int t0 = ((n - 1) / kLmsChunk) * kLmsChunk;
    if (n > 0 && t0 + l < n) pio[t0 + l] = L.pout[l];
  });
}

}  // namespace sacamd
