// oracle/sac_oracle.cpp -- TEST INFRASTRUCTURE ONLY (checker; never linked by sac_amd/).
//
// A from-scratch CPU restatement of the Sac v0.7.25 encode hot path (and the matching decode
// path needed to prove losslessness), following the reference algorithm file:line by
// file:line.  Citations are relative to /root/reference/src.
//
// Parity status: PINNED.  Every stage is checked in tests/ against oracle/_ref (the genuine
// reference classes compiled from /root/reference, see oracle/ref_driver.cpp) and against
// golden vectors generated from it (tests/golden/, generator tests/golden/make_golden.py).
//
// Floating point: this file is compiled with -ffp-contract=off and every fused multiply-add the
// reference *binary* contains (g++ 11 -O3 -mfma, default -ffp-contract=fast, plus the
// hand-written AVX2 intrinsics of common/math.h) is written as an explicit fma() here, so the
// arithmetic is a compiler-independent specification that the HIP kernels mirror.
#include "sac_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#define API extern "C" __attribute__((visibility("default")))

namespace orc {

// =====================================================================================
// profile (libsac/profile.cpp:3-89) and parameter mapping (libsac/libsac.cpp:37-92)
// =====================================================================================
struct Coef { float vmin, vmax, vdef; };
struct Profile {
  Coef c[58];
  void load_default() {
    const int mo_lpc = 32, wb = 13;
    auto S = [&](int i, double a, double b, double d) {
      c[i].vmin = (float)a; c[i].vmax = (float)b; c[i].vdef = (float)d;
    };
    for (auto &e : c) e = {0.f, 0.f, 0.f};
    S(0, 0.99, 0.9999, 0.998); S(1, 1.0, 100.0, 25.0);
    S(2, 0.001, 1.0, 0.1); S(3, 0.001, 1.0, 0.12); S(4, 0.001, 1.0, 0.06); S(5, 0.001, 1.0, 0.04);
    S(6, 0.98, 1, 1.0); S(7, 0.0, 1.0, 0.8); S(8, 0.0, 1.0, 0.8);
    S(10, 0.0005, 0.05, 0.005); S(11, 0.8, 0.9999, 0.95);
    S(12, 0.99, 0.9999, 0.998); S(13, 1.0, 100.0, 25.0);
    S(14, 0.001, 1.0, 0.1); S(15, 0.001, 1.0, 0.12); S(16, 0.001, 1.0, 0.06); S(17, 0.001, 1.0, 0.04);
    S(18, 0.98, 1, 1.0); S(19, 0.0, 1.0, 0.8); S(20, 0.0, 1.0, 0.8); S(21, 0.0, 1.0, 0.8);
    S(22, 0.0005, 0.05, 0.005); S(23, 0.8, 0.9999, 0.95);
    S(24, 4, mo_lpc, 16); S(25, 4, mo_lpc, 16); S(26, 0, mo_lpc, 8); S(27, -mo_lpc, mo_lpc, 8);
    S(9, 0, mo_lpc, 0);
    S(28, 256, 1 << wb, 1280); S(29, 32, 1 << (wb - 1), 256); S(30, 4, 1 << (wb - 2), 32);
    S(31, 256, 1 << wb, 1280); S(32, 32, 1 << (wb - 1), 256); S(33, 4, 1 << (wb - 2), 32);
    S(34, 0, 1, 0.5); S(35, 0.1, 2, 0.8); S(36, 0.1, 10, 2);
    S(53, 0, 1, 0.5); S(54, 0.1, 2, 0.8); S(55, 0.1, 10, 2);
    S(56, 0.0, 0.5, 0.1); S(57, 0.0, 0.5, 0.1);
    S(37, 2, 1 << (wb - 3), 4); S(38, 2, 1 << (wb - 3), 4);
    S(39, 0.98, 1, 1.0); S(40, 0.98, 1, 1.0);
    S(41, 1, 10, 4); S(42, 0.1, 10.0, 5);
    S(43, 0.001, 0.005, 0.0015); S(44, 0.001, 0.005, 0.0015);
    S(45, 4, 10, 5);
    S(46, 0.98, 1, 1.0); S(48, 0.98, 1, 1.0); S(50, 0.0, 1.0, 0.8);
    S(47, 0.98, 1, 1.0); S(49, 0.98, 1, 1.0); S(51, 0.0, 1.0, 0.8); S(52, 0.0, 1.0, 0.8);
  }
};

struct ChanParam {           // everything one channel's predictor needs
  int n_ols;                 // regressor length
  int k;
  double lambda, ols_nu, beta_sum, beta_pow, beta_add;
  int vn[4];
  double vmu[4], vmudecay[4], vpowdecay[4];
  double mu_mix, mu_mix_beta;
  int lm_n;
  double lm_alpha, proj_alpha;
  double bias_mu;
  int bias_scale;
};
struct Params {
  int nA, nB, nM0, nS0, nS1, k, ch_ref;
  ChanParam ch[2];
};

static inline int iround(float v) { return (int)std::round((double)v); }
// `(int32_t)std::round(pd)` of libsac.cpp:106 / :154 as the reference's x86-64 build executes it: cvttsd2si returns INT_MIN for
// values outside the int32 range and for NaN (C++ leaves the conversion undefined; 24-bit material does overshoot past 2^31
// in the first samples of a frame, and the decoder repeats whatever the encoder's conversion gave)
static inline int32_t cvt_i32_x86(double r) { return (r >= -2147483648.0 && r < 2147483648.0) ? (int32_t)r : INT32_MIN; }

// libsac.cpp:37-92
static void set_param(Params &p, const float *g, bool optimize, int optk) {
  p.k = optimize ? optk : 1;
  ChanParam &a = p.ch[0], &b = p.ch[1];
  a.k = b.k = p.k;
  a.lambda = g[0]; a.ols_nu = g[1];
  a.vn[0] = iround(g[28]); a.vn[1] = iround(g[29]); a.vn[2] = iround(g[30]); a.vn[3] = iround(g[37]);
  b.vn[0] = iround(g[31]); b.vn[1] = iround(g[32]); b.vn[2] = iround(g[33]); b.vn[3] = iround(g[38]);
  const int mu0[4] = {2, 3, 4, 5}, mu1[4] = {14, 15, 16, 17};
  for (int i = 0; i < 4; i++) {
    a.vmu[i] = (double)g[mu0[i]] / double(a.vn[i]);
    b.vmu[i] = (double)g[mu1[i]] / double(b.vn[i]);
  }
  a.vmudecay[0] = g[6]; a.vmudecay[1] = g[39]; a.vmudecay[2] = g[46]; a.vmudecay[3] = g[47];
  a.vpowdecay[0] = g[7]; a.vpowdecay[1] = g[8]; a.vpowdecay[2] = g[50]; a.vpowdecay[3] = g[51];
  a.mu_mix = g[10]; a.mu_mix_beta = g[11];
  b.lambda = g[12]; b.ols_nu = g[13];
  b.vmudecay[0] = g[18]; b.vmudecay[1] = g[40]; b.vmudecay[2] = g[48]; b.vmudecay[3] = g[49];
  b.vpowdecay[0] = g[19]; b.vpowdecay[1] = g[20]; b.vpowdecay[2] = g[21]; b.vpowdecay[3] = g[52];
  b.mu_mix = g[22]; b.mu_mix_beta = g[23];
  p.nA = iround(g[24]); p.nB = iround(g[25]); p.nS0 = iround(g[26]); p.nS1 = iround(g[27]);
  p.nM0 = iround(g[9]);
  a.beta_sum = g[34]; a.beta_pow = g[35]; a.beta_add = g[36];
  b.beta_sum = g[53]; b.beta_pow = g[54]; b.beta_add = g[55];
  a.proj_alpha = g[56]; b.proj_alpha = g[57];
  a.lm_n = b.lm_n = iround(g[41]);
  a.lm_alpha = b.lm_alpha = g[42];
  a.bias_mu = g[43]; b.bias_mu = g[44];
  a.bias_scale = b.bias_scale = iround(g[45]);
  p.ch_ref = 0;
  if (p.nS1 < 0) { p.nS1 = -p.nS1; p.ch_ref = 1; }
  a.n_ols = p.nA + p.nM0;
  b.n_ols = p.nB + p.nS0 + p.nS1;
}

// =====================================================================================
// inner products in the reference's exact summation order
// =====================================================================================
// How g++ 11 -O3 -mavx2 -mfma compiles the reference's scalar reduction loops (established from
// the disassembly of oracle/_ref, see DESIGN.md "canonical arithmetic"): the vectoriser runs
// before FMA contraction and keeps fp reductions in order, so a loop `acc (+|-)= a[k]*b[k]`
// becomes: full groups of 4 and then one pair with the products rounded separately
// (vmulpd + scalar vaddsd/vsubsd chain), and only a final odd element as a fused multiply-add.
template <class FP>
static inline double fold_add(double acc, size_t m, FP prod_ab) {
  // prod_ab(k, &a, &b): the k-th term is a*b (a may itself be a rounded product)
  size_t k = 0; double a, b;
  for (; k + 4 <= m; k += 4) for (int u = 0; u < 4; u++) { prod_ab(k + u, a, b); acc = acc + a * b; }
  if (m - k >= 2) { prod_ab(k, a, b); acc = acc + a * b; prod_ab(k + 1, a, b); acc = acc + a * b; k += 2; }
  if (k < m) { prod_ab(k, a, b); acc = std::fma(a, b, acc); }
  return acc;
}
template <class FP>
static inline double fold_sub(double acc, size_t m, FP prod_ab) {
  size_t k = 0; double a, b;
  for (; k + 4 <= m; k += 4) for (int u = 0; u < 4; u++) { prod_ab(k + u, a, b); acc = acc - a * b; }
  if (m - k >= 2) { prod_ab(k, a, b); acc = acc - a * b; prod_ab(k + 1, a, b); acc = acc - a * b; k += 2; }
  if (k < m) { prod_ab(k, a, b); acc = std::fma(-a, b, acc); }
  return acc;
}

// libstdc++ std::transform_reduce(first1,last1,first2,0.0) (/usr/include/c++/11/numeric:380-396)
// as compiled: 4-groups v1=fma(a1,b1,a0*b0), v2=fma(a3,b3,a2*b2), init+=(v1+v2); the <4 tail is
// the vectorised sequential loop above (pair unfused, odd last fused).
static inline double tr_dot(const double *a, const double *b, size_t n) {
  double init = 0.0;
  while (n >= 4) {
    double v1 = std::fma(a[1], b[1], a[0] * b[0]);
    double v2 = std::fma(a[3], b[3], a[2] * b[2]);
    init = init + (v1 + v2);
    a += 4; b += 4; n -= 4;
  }
  return fold_add(init, n, [&](size_t k, double &u, double &v) { u = a[k]; v = b[k]; });
}

// slmath::dot, common/math.h:130-161
static inline double dot_ref(const double *x, const double *y, size_t n) {
  double total = 0.0;
  size_t i = 0;
  if (n >= 8) {
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (; i + 8 <= n; i += 8)
      for (int c = 0; c < 4; c++) {
        s1[c] = std::fma(x[i + c], y[i + c], s1[c]);
        s2[c] = std::fma(x[i + 4 + c], y[i + 4 + c], s2[c]);
      }
    for (int c = 0; c < 4; c++) s1[c] = s1[c] + s2[c];
    total = ((s1[0] + s1[1]) + s1[2]) + s1[3];
  }
  total += tr_dot(x + i, y + i, n - i);
  return total;
}

// slmath::calc_s2pow, common/math.h:164-191
static inline double s2pow_ref(const double *x, const double *pw, size_t n) {
  double spow = 0.0;
  size_t i = 0;
  if (n >= 8) {
    double s[4] = {0, 0, 0, 0};
    for (; i + 4 <= n; i += 4)
      for (int c = 0; c < 4; c++) s[c] = std::fma(pw[i + c], x[i + c] * x[i + c], s[c]);
    spow = ((s[0] + s[1]) + s[2]) + s[3];
  }
  // transform_reduce with op2 = p*(x*x)
  double init = 0.0;
  size_t m = n - i;
  const double *a = x + i, *b = pw + i;
  while (m >= 4) {
    double v1 = std::fma(a[1] * a[1], b[1], (a[0] * a[0]) * b[0]);
    double v2 = std::fma(a[3] * a[3], b[3], (a[2] * a[2]) * b[2]);
    init = init + (v1 + v2);
    a += 4; b += 4; m -= 4;
  }
  init = fold_add(init, m, [&](size_t k, double &u, double &v) { u = a[k] * a[k]; v = b[k]; });
  spow += init;
  return spow;
}

// =====================================================================================
// LDL^T (common/math.h:14-78) and OLS (pred/ols.cpp:7-57)
// =====================================================================================
struct LDLT {
  int n;
  std::vector<double> L, D, invD, y, z;
  void init(int n_) { n = n_; L.assign((size_t)n * n, 0.0); D.assign(n, 0.0); invD.assign(n, 0.0); y.assign(n, 0.0); z.assign(n, 0.0); }
  bool factor(const double *A, double nu) {
    const double eps = 1e-12;
    for (int j = 0; j < n; ++j) {
      double dj = A[j * n + j] + nu;
      dj = fold_sub(dj, j, [&](size_t k, double &u, double &v) { u = L[j * n + k] * L[j * n + k]; v = D[k]; });
      if (dj < eps) return false;
      const double invDj = 1.0 / dj;
      D[j] = dj; invD[j] = invDj;
      for (int i = j + 1; i < n; ++i) {
        double lij = A[i * n + j];
        lij = fold_sub(lij, j, [&](size_t k, double &u, double &v) { u = L[i * n + k] * L[j * n + k]; v = D[k]; });
        L[i * n + j] = lij * invDj;
      }
    }
    return true;
  }
  void solve(const double *b, double *x) {
    for (int i = 0; i < n; ++i) {
      double s = b[i];
      s = fold_sub(s, i, [&](size_t k, double &u, double &v) { u = L[i * n + k]; v = y[k]; });
      y[i] = s;
    }
    for (int i = 0; i < n; ++i) z[i] = y[i] * invD[i];
    for (int i = n - 1; i >= 0; --i) {
      double s = z[i];
      for (int k = i + 1; k < n; ++k) s = std::fma(-L[k * n + i], x[k], s);
      x[i] = s;
    }
  }
};

struct OLS {
  int n, kmax, km;
  double lambda, nu, beta_pow, beta_add, beta_sum, esum, pred;
  std::vector<double> x, w, b, mcov;
  LDLT ldlt;
  void init(int n_, int kmax_, double lambda_, double nu_, double bsum, double bpow, double badd) {
    n = n_; kmax = kmax_; lambda = lambda_; nu = (1.0 - lambda_) * nu_;
    beta_pow = bpow; beta_add = badd; beta_sum = bsum; esum = 0.0; pred = 0.0; km = 0;
    x.assign(n, 0.0); w.assign(n, 0.0); b.assign(n, 0.0); mcov.assign((size_t)n * n, 0.0);
    ldlt.init(n);
  }
  double predict() { return (pred = dot_ref(x.data(), w.data(), n)); }
  void update(double val) {
    const double e = val - pred;
    esum = std::fma(beta_sum, esum, std::fabs(e));             // RunSumGEO, utils.h:50-51
    const double c = std::pow(esum + beta_add, -beta_pow);
    const double ff = (1.0 - lambda) * c;
    for (int j = 0; j < n; j++) {
      const double xj = x[j];
      for (int i = 0; i <= j; i++) mcov[j * n + i] = std::fma(lambda, mcov[j * n + i], ff * (xj * x[i]));
      b[j] = std::fma(lambda, b[j], ff * (xj * val));
    }
    km++;
    if (km >= kmax) {
      if (ldlt.factor(mcov.data(), nu)) ldlt.solve(b.data(), w.data());
      km = 0;
    }
  }
};

// =====================================================================================
// NLMS stage (pred/ls.h:10-62), history = index 0 newest (common/histbuf.h:61-88)
// =====================================================================================
struct NLMS {
  int n;
  double mu, sum_powtab, pred;
  std::vector<double> hist;   // contiguous window, [0] newest
  std::vector<double> w, mutab, powtab;
  void init(int n_, double mu_, double mu_decay, double pow_decay) {
    n = n_; mu = mu_; pred = 0.0;
    hist.assign(n, 0.0); w.assign(n, 0.0); mutab.assign(n, 0.0); powtab.assign(n, 0.0);
    sum_powtab = 0;
    for (int i = 0; i < n; i++) {
      powtab[i] = 1.0 / (std::pow((double)(1 + i), pow_decay));
      sum_powtab += powtab[i];
      mutab[i] = std::pow(mu_decay, (double)i);
    }
  }
  double predict() { return (pred = dot_ref(hist.data(), w.data(), n)); }
  void update(double val) {
    const double spow = s2pow_ref(hist.data(), powtab.data(), n);
    const double wgrad = mu * (val - pred) * sum_powtab / (spow + 1.0);
    for (int i = 0; i < n; i++) {
      double v = std::fma(mutab[i], wgrad * hist[i], w[i]);
      w[i] = std::min(std::max(v, -10.0), 10.0);     // std::clamp(w,-10,10)
    }
    if (n > 1) std::memmove(&hist[1], &hist[0], (size_t)(n - 1) * sizeof(double));
    hist[0] = val;
  }
};

// =====================================================================================
// mixer experts LS_ADA<L1|L2,Uniform> (pred/ls.h:214-241), BlendExp (pred/blend.h),
// BlendLS (pred/cascade.h:11-57)
// =====================================================================================
static inline double sgnd(double x) { return (double)((x > 0) - (x < 0)); }

struct LSAda {
  int n; bool l1;
  double mu, beta, beta1;
  std::vector<double> w, eg;
  void init(int n_, double mu_, double beta_, bool l1_) {
    n = n_; mu = mu_; beta = beta_; beta1 = 1.0 - beta_; l1 = l1_;
    w.assign(n, 1.0 / n); eg.assign(n, 0.0);
  }
  double predict(const double *x) const { return dot_ref(x, w.data(), n); }
  void update(const double *x, double error) {
    const double loss = l1 ? sgnd(error) : error;
    for (int i = 0; i < n; ++i) {
      const double grad = loss * x[i];
      eg[i] = std::fma(beta, eg[i], beta1 * grad * grad);
      const double mu_scaled = mu / (std::sqrt(eg[i]) + 1e-5);
      w[i] = std::fma(mu_scaled, grad, w[i]);
    }
  }
};

struct Blend2 {              // BlendExp<RunSumEMA>(2, 0.95, 1.0)
  double x[2], w[2], rs[2], px;
  void init() { w[0] = w[1] = 0.5; rs[0] = rs[1] = 0.0; x[0] = x[1] = 0.0; px = 0.0; }
  double predict(const double *in) { x[0] = in[0]; x[1] = in[1]; return (px = dot_ref(x, w, 2)); }
  void update(double target) {
    const double alpha = 0.95, beta = 1.0;
    for (int i = 0; i < 2; i++) {
      const double loss = std::abs(target - x[i]);
      rs[i] = std::fma(alpha, rs[i], (1.0 - alpha) * (-loss));   // RunSumEMA, utils.h:48-49
    }
    double zm[2], maxz = -std::numeric_limits<double>::infinity();
    for (int i = 0; i < 2; i++) { zm[i] = beta * rs[i]; maxz = std::max(maxz, zm[i]); }
    double total = 0.0;
    for (int i = 0; i < 2; i++) { w[i] = std::exp(zm[i] - maxz); total += w[i]; }
    const double inv = 1.0 / total;
    for (int i = 0; i < 2; i++) w[i] *= inv;
  }
};

// =====================================================================================
// RLS with adaptive lambda (pred/rls.h, pred/rls.cpp)
// =====================================================================================
struct RLS {
  int n; double px, gamma, beta, S0, S1;
  std::vector<double> x, w, P, ph;
  void init(int n_, double gamma_, double beta_) {
    n = n_; gamma = gamma_; beta = beta_; S0 = S1 = 0.0; px = 0.0;
    x.assign(n, 0.0); w.assign(n, 0.0); P.assign((size_t)n * n, 0.0); ph.assign(n, 0.0);
    for (int i = 0; i < n; i++) P[i * n + i] = 1.0;
  }
  double predict() { return (px = dot_ref(x.data(), w.data(), n)); }
  void update_hist(double val) {
    const double err = val - px;
    for (int i = 0; i < n; i++) ph[i] = dot_ref(&P[i * n], x.data(), n);
    const double phi = std::max(dot_ref(x.data(), ph.data(), n), 1e-8);
    const double err2 = err * err;
    const double R = std::max(S0 - S1, 1e-5);
    const double nis = err2 / (phi + R);
    const double m = std::exp(-gamma * nis);
    const double lmin = 0.99, lmax = 0.999;
    const double alpha = std::fma(lmax - lmin, m, lmin);
    const double denom = 1. / (alpha + phi);
    const double inv_alpha = 1.0 / alpha;
    for (int i = 0; i < n; i++)
      for (int j = 0; j <= i; j++) {
        const double mm = ph[i] * ph[j];
        const double v = std::fma(-denom, mm, P[i * n + j]) * inv_alpha;
        P[i * n + j] = P[j * n + i] = v;
      }
    for (int i = 0; i < n; i++) w[i] = std::fma(err, denom * ph[i], w[i]);
    S0 = std::fma(beta, S0, (1.0 - beta) * err2);
    S1 = std::fma(beta, S1, (1.0 - beta) * phi);
    if (n > 1) std::memmove(&x[1], &x[0], (size_t)(n - 1) * sizeof(double));
    if (n) x[0] = val;
  }
};

// =====================================================================================
// Cascade (pred/cascade.h:75-131)
// =====================================================================================
struct Cascade {
  double lo, hi, p_alpha, pred;
  NLMS st[4];
  RLS lm;
  LSAda ex[2];
  Blend2 sm;
  double p[5], bp[5], ep[2];
  void init(int lo_, int hi_, const ChanParam &c) {
    lo = lo_; hi = hi_; p_alpha = c.proj_alpha; pred = 0.0;
    for (int i = 0; i < 4; i++) st[i].init(c.vn[i], c.vmu[i], c.vmudecay[i], c.vpowdecay[i]);
    lm.init(c.lm_n, c.lm_alpha, 0.95);
    ex[0].init(5, c.mu_mix, c.mu_mix_beta, true);
    ex[1].init(5, c.mu_mix, c.mu_mix_beta, false);
    sm.init();
    for (int i = 0; i < 5; i++) p[i] = bp[i] = 0.0;
  }
  double predict() {
    for (int i = 0; i < 4; i++) p[i] = st[i].predict();
    p[4] = lm.predict();
    for (int e = 0; e < 2; e++) ep[e] = ex[e].predict(p);
    return (pred = sm.predict(ep));
  }
  void update(double target) {
    double p_prefix = 0.0;
    for (int i = 0; i <= 4; i++) {
      double ew[2] = {ex[0].w[i], ex[1].w[i]};
      const double w = std::max(dot_ref(ew, sm.w, 2), 0.0);
      const double px = std::fma(1.0 - p_alpha, p_prefix, p_alpha * pred);
      bp[i] = target - std::min(std::max(px, lo), hi);
      p_prefix = std::fma(w, p[i], p_prefix);
    }
    for (int i = 0; i < 4; i++) st[i].update(bp[i]);
    lm.update_hist(bp[4]);
    for (int e = 0; e < 2; e++) ex[e].update(p, target - ep[e]);
    sm.update(target);
  }
};

// =====================================================================================
// Bias estimator (pred/bias.h:16-175)
// =====================================================================================
struct Bias {
  struct Cnt { double cnt, val; };
  Cnt c0[64], c1[64], c2[64];
  double mixw[4][3];
  double hin[8], hdl[8], pt[3];
  int ctx0, ctx1, ctx2, mix_ctx, nscale;
  double px, pbias, mu, mean, var;
  void init(double mu_, int nb_scale) {
    mu = mu_; nscale = 1 << nb_scale;
    for (int i = 0; i < 64; i++) c0[i] = c1[i] = c2[i] = {4.0, 0.0};
    std::memset(mixw, 0, sizeof(mixw)); std::memset(hin, 0, sizeof(hin)); std::memset(hdl, 0, sizeof(hdl));
    ctx0 = ctx1 = ctx2 = mix_ctx = 0; px = pbias = 0.0; mean = var = 0.0; pt[0] = pt[1] = pt[2] = 0;
  }
  void calc_ctx(double p) {
    int b0 = hin[0] > p ? 0 : 1;
    int b2 = hdl[0] < 0 ? 0 : 1, b3 = hdl[1] < 0 ? 0 : 1, b4 = hdl[2] < 0 ? 0 : 1;
    int b5 = hdl[1] < hdl[0] ? 0 : 1, b6 = hdl[2] < hdl[1] ? 0 : 1, b7 = hdl[3] < hdl[2] ? 0 : 1, b8 = hdl[4] < hdl[3] ? 0 : 1;
    int b9 = (std::fabs(hdl[0])) > 32 ? 0 : 1;
    int b10 = 2 * hin[0] - hin[1] > p ? 0 : 1;
    int b11 = 3 * hin[0] - 3 * hin[1] + hin[2] > p ? 0 : 1;
    double sum = 0;
    for (int i = 0; i < 5; i++) sum += std::fabs(hdl[i]);
    sum /= 5.0;
    int t = 0;
    if (sum > 512) t = 2; else if (sum > 32) t = 1;
    ctx0 = b0 + (b2 << 1) + (b9 << 2) + (b10 << 3) + (b11 << 4);
    ctx1 = b2 + (b3 << 1) + (b4 << 2);
    ctx2 = b5 + (b6 << 1) + (b7 << 2) + (b8 << 3);
    mix_ctx = t;
  }
  double predict(double pred) {
    px = pred;
    calc_ctx(pred);
    pt[0] = c0[ctx0].val / c0[ctx0].cnt;
    pt[1] = c1[ctx1].val / c1[ctx1].cnt;
    pt[2] = c2[ctx2].val / c2[ctx2].cnt;
    pbias = dot_ref(pt, mixw[mix_ctx], 3);
    return px + pbias;
  }
  void cupd(Cnt &c, double delta, double w) {
    c.val = c.val + w * delta;   // not contracted in the reference binary (SLP-vectorised pair)
    c.cnt += w;
    if (c.cnt >= nscale) { c.val *= 0.5; c.cnt *= 0.5; }
  }
  void update(double val) {
    const double delta = val - std::round(px);
    std::memmove(&hin[1], &hin[0], 7 * sizeof(double)); hin[0] = val;
    std::memmove(&hdl[1], &hdl[0], 7 * sizeof(double)); hdl[0] = delta;
    const double v = std::max(0.0, var);
    const double diff = delta - mean;
    const double z = diff * diff / (v + 1E-5);
    const double w = std::exp(-0.5 * z);
    cupd(c0[ctx0], delta, w); cupd(c1[ctx1], delta, w); cupd(c2[ctx2], delta, w);
    // RunMeanVar(0.998), utils.h:75-110
    const double a = 0.998;
    const double old_mean = mean;
    mean = std::fma(a, mean, (1.0 - a) * delta);
    var = std::fma(a, var, (1.0 - a) * ((delta - old_mean) * (delta - mean)));
    // SSLMS, ls.h:279-292
    const double wf = mu * sgnd(delta - pbias);
    for (int i = 0; i < 3; i++) mixw[mix_ctx][i] = std::fma(wf, sgnd(pt[i]), mixw[mix_ctx][i]);
  }
};

// =====================================================================================
// Predictor (libsac/pred.h, pred.cpp) and the frame loop (libsac.cpp:94-142)
// =====================================================================================
struct ChanPred {
  OLS ols; Cascade lms; Bias be;
  double p_lpc, p_lms;
  void init(int lo, int hi, const ChanParam &c) {
    ols.init(c.n_ols, c.k, c.lambda, c.ols_nu, c.beta_sum, c.beta_pow, c.beta_add);
    lms.init(lo, hi, c);
    be.init(c.bias_mu, c.bias_scale);
    p_lpc = p_lms = 0.0;
  }
  double predict() { p_lpc = ols.predict(); p_lms = lms.predict(); return be.predict(p_lpc + p_lms); }
  void update(double val) { ols.update(val); lms.update(val - p_lpc); be.update(val); }
};

struct Predictor {
  Params P; ChanPred c[2];
  void init(const Params &p, const int32_t lo[2], const int32_t hi[2]) {
    P = p;
    // quirk: Cascade ranges are r0=framestats[0], r1=framestats[1] regardless of ch_ref
    c[0].init(lo[0], hi[0], p.ch[0]);
    c[1].init(lo[1], hi[1], p.ch[1]);
  }
  void fill0(const int32_t *s0, int i0, const int32_t *s1, int i1) {
    double *b = c[0].ols.x.data(); int bp = 0;
    for (int i = i0 - P.nA; i < i0; i++) b[bp++] = (i >= 0) ? s0[i] : 0.0;
    for (int i = i1 - P.nM0; i < i1; i++) b[bp++] = (i >= 0) ? s1[i] : 0.0;
  }
  void fill1(const int32_t *s0, const int32_t *s1, int i1, int ns) {
    double *b = c[1].ols.x.data(); int bp = 0;
    for (int i = i1 - P.nB; i < i1; i++) b[bp++] = (i >= 0) ? s1[i] : 0.0;
    for (int i = i1 - P.nS0; i < i1 + P.nS1; i++) b[bp++] = (i >= 0 && i < ns) ? s0[i] : 0.0;
  }
};

struct Trace { double *pd, *plpc, *plms; };

// samples planar [nch][total], mean removed. stats per channel {min,max,mean}.
static void predict_frame(int nch, int total, const int32_t *samples, const int32_t *stats,
                          const float *coefs, int from, int n, bool optimize, int optk,
                          int32_t *error, int32_t *pred, Trace *tr) {
  Params P; set_param(P, coefs, optimize, optk);
  int32_t lo[2], hi[2], mean[2];
  for (int ch = 0; ch < 2; ch++) {
    int s = ch < nch ? ch : 0;
    lo[ch] = stats[3 * s]; hi[ch] = stats[3 * s + 1]; mean[ch] = stats[3 * s + 2];
  }
  Predictor pr; pr.init(P, lo, hi);
  auto step = [&](int chp, int ch, int32_t val, int idx) {
    double pd = pr.c[chp].predict();
    int32_t pi = std::min(std::max(cvt_i32_x86(std::round(pd)), lo[ch]), hi[ch]);
    if (tr) {
      tr->pd[(size_t)ch * n + idx] = pd;
      tr->plpc[(size_t)ch * n + idx] = pr.c[chp].p_lpc;
      tr->plms[(size_t)ch * n + idx] = pr.c[chp].p_lms;
    }
    if (pred) pred[(size_t)ch * n + idx] = pi + mean[ch];
    error[(size_t)ch * n + idx] = val - pi;
    pr.c[chp].update(val);
  };
  if (nch == 1) {
    const int32_t *src = samples + from;
    for (int idx = 0; idx < n; idx++) { pr.fill0(src, idx, src, idx); step(0, 0, src[idx], idx); }
  } else {
    int ch0 = P.ch_ref, ch1 = 1 - ch0;
    const int32_t *s0 = samples + (size_t)ch0 * total + from, *s1 = samples + (size_t)ch1 * total + from;
    int i0 = 0, i1 = 0;
    while (i0 < n || i1 < n) {
      if (i0 < n) { pr.fill0(s0, i0, s1, i1); step(0, ch0, s0[i0], i0); i0++; }
      if (i0 >= P.nS1) { pr.fill1(s0, s1, i1, n); step(1, ch1, s1[i1], i1); i1++; }
    }
  }
}

// =====================================================================================
// integer helpers (common/utils.h:248-267)
// =====================================================================================
static inline int32_t s2u(int32_t v) { if (v < 0) return 2 * (-v); if (v > 0) return 2 * v - 1; return v; }
static inline int32_t u2s(int32_t v) { return (v & 1) ? ((v + 1) >> 1) : -(v >> 1); }
static inline int ilog2(int v) { int nb = 0; while (v >>= 1) nb++; return nb; }

// =====================================================================================
// model primitives (model/*.h) -- integer exact
// =====================================================================================
enum { PBITS = 15, PSCALE = 1 << 15, PSCALEm = PSCALE - 1 };

struct Domain {               // model/domain.h
  int fwd[PSCALE]; int inv[4095]; int dmin_ = -2047, dmax_ = 2047; int mn, mx;
  Domain() {
    for (int i = 0; i < PSCALE; i++) {
      double f = std::max(i, 1) / (double)PSCALE;
      fwd[i] = (int)std::round(std::log(f / (1.0 - f)) * 256);
    }
    mn = fwd[0]; mx = fwd[PSCALE - 1];
    for (int i = -2047; i <= 2047; i++) inv[i + 2047] = (int)std::round(PSCALE / (1.0 + std::exp(-double(i) / 256.0)));
  }
  int Fwd(int p) const { return fwd[p]; }
  int Inv(int x) const { if (x < -2047) return 1; if (x > 2047) return PSCALEm; return inv[x + 2047]; }
};
static const Domain &dom() { static Domain d; return d; }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int idiv_s(int val, int s) { return val < 0 ? -(((-val) + (1 << (s - 1))) >> s) : (val + (1 << (s - 1))) >> s; }
static inline int idiv_s64(int64_t val, int s) { return (int)(val < 0 ? -(((-val) + (1 << (s - 1))) >> s) : (val + (1 << (s - 1))) >> s); }

struct CntLimit {             // LinearCounterLimit, counter.h:53-69
  uint16_t p1 = PSCALE >> 1, counter = 0;
  void update(int bit, int limit) {
    if (counter < limit) counter++;
    const int d = PSCALE / (counter + 3);
    int dp = bit ? ((PSCALE - p1) * d) >> PBITS : -((p1 * d) >> PBITS);
    p1 = (uint16_t)clampi(p1 + dp, 1, PSCALEm);
  }
};
struct Cnt16 {                // LinearCounter16::update(bit,L), counter.h:31-37
  uint16_t p1 = PSCALE >> 1;
  void update(int bit, int L) {
    int err = (bit << PBITS) - p1;
    int px = int(p1) + idiv_s(L * err, PBITS);
    p1 = (uint16_t)clampi(px, 1, PSCALEm);
  }
};
struct Mixer {                // NMixLogistic, mixer.h:59-101
  int n = 0; int w[5] = {0, 0, 0, 0, 0}; int16_t x[5] = {0, 0, 0, 0, 0}; int16_t pd = 0;
  int predict(const int *p) {
    int64_t sum = 0;
    for (int i = 0; i < n; i++) { x[i] = (int16_t)dom().Fwd(p[i]); sum += int64_t(w[i] * x[i]); }
    int s = idiv_s64(sum, 16);
    pd = (int16_t)clampi(dom().Inv(s), 1, PSCALEm);
    return pd;
  }
  void update(int bit, int rate) {
    int err = (bit << PBITS) - pd;
    for (int i = 0; i < n; i++) {
      int de = idiv_s(x[i] * err, 12);
      int wd = idiv_s(de * rate, 12);
      w[i] = clampi(w[i] + wd, -(1 << 19), (1 << 19) - 1);
    }
  }
};
template <int N> struct SSENL {   // sse.h:84-125
  int tscale, xscale, lb = 0; uint16_t p_quant = 0;
  Cnt16 map[2][N + 1];
  SSENL() {
    tscale = dom().mx; xscale = (2 * tscale) / (N - 1); if (xscale == 0) xscale = 1;
    for (int i = 0; i <= N; i++) { int x = dom().Inv(i * xscale - tscale); map[0][i].p1 = (uint16_t)x; map[1][i].p1 = (uint16_t)x; }
  }
  int predict(int p1) {
    int pq = std::min(2 * tscale, std::max(0, dom().Fwd(p1) + tscale));
    p_quant = (uint16_t)(pq / xscale);
    int p_mod = pq - (p_quant * xscale);
    int pl = map[lb][p_quant].p1, ph = map[lb][p_quant + 1].p1;
    int px = (pl * (xscale - p_mod) + ph * p_mod) / xscale;
    return clampi(px, 1, PSCALEm);
  }
  void update(int bit, int rate) { map[lb][p_quant].update(bit, rate); map[lb][p_quant + 1].update(bit, rate); lb = bit; }
};

// RangeCoderSH (model/range.cpp:54-92)
struct RangeEnc {
  std::vector<uint8_t> out; uint32_t range = 0xFFFFFFFFu, FFNum = 0, Cache = 0; uint64_t lowc = 0;
  void shift_low() {
    uint32_t Carry = uint32_t(lowc >> 32), low = uint32_t(lowc);
    if (low < 0xFF000000u || Carry) {
      out.push_back((uint8_t)(Cache + Carry));
      for (; FFNum != 0; FFNum--) out.push_back((uint8_t)(Carry - 1));
      Cache = low >> 24;
    } else FFNum++;
    lowc = (uint64_t)(uint32_t)(low << 8);
  }
  void encode(uint32_t p1, int bit) {
    const uint32_t rnew = (uint32_t)((uint64_t(range) * ((uint32_t)(PSCALE - p1) << (32 - PBITS))) >> 32);
    if (bit) { range -= rnew; lowc += rnew; } else range = rnew;
    while (range < 0x01000000u) { range <<= 8; shift_low(); }
  }
  void stop() { for (int i = 0; i < 5; i++) shift_low(); }
};
struct RangeDec {
  const uint8_t *in; size_t len, pos = 0; uint32_t range = 0xFFFFFFFFu, code = 0;
  int get() { return pos < len ? in[pos++] : -1; }
  RangeDec(const uint8_t *p, size_t l) : in(p), len(l) { for (int i = 0; i < 5; i++) code = (code << 8) + get(); }
  int decode(uint32_t p1) {
    const uint32_t rnew = (uint32_t)((uint64_t(range) * ((uint32_t)(PSCALE - p1) << (32 - PBITS))) >> 32);
    int bit = (code >= rnew);
    if (bit) { range -= rnew; code -= rnew; } else range = rnew;
    while (range < 0x01000000u) { range <<= 8; code = (code << 8) + get(); }
    return bit;
  }
};

// =====================================================================================
// BitplaneCoder (libsac/vle.cpp) -- shared model for encode/decode
// =====================================================================================
struct Bitplane {
  int maxbpn, numsamples, bpn = 0, sample = 0, pestimate = 0;
  uint32_t state = 0;
  std::vector<CntLimit> csig0, csig1, cref0, cref1, cref2, cref3;
  CntLimit p_laplace[32];
  Mixer lmixref[32], lmixsig[128], ssemix;
  std::vector<SSENL<15>> sse;
  std::vector<int> msb;
  int sigst[17];
  int *pabuf = nullptr;
  CntLimit *pl, *pc1, *pc2, *pc3, *pc4; Mixer *plmix; SSENL<15> *ps1, *ps2;

  Bitplane(int maxbpn_, int n) : maxbpn(maxbpn_), numsamples(n), csig0(1 << 16), csig1(80), cref0(32), cref1(256), cref2(64), cref3(256), sse(160), msb(n, 0) {
    for (auto &m : lmixref) m.n = 5;
    for (auto &m : lmixsig) m.n = 3;
    ssemix.n = 2;
    const double theta = 0.99;
    for (int i = 0; i < 32; i++) {
      double pw = (i < 31) ? std::pow(theta, (double)(1 << i)) : std::pow(theta, (double)std::numeric_limits<int>::min());
      int p = std::min(std::max((int)std::round((1.0 - 1.0 / (1 + pw)) * PSCALE), 1), (int)PSCALEm);
      p_laplace[i].p1 = (uint16_t)p;
    }
  }
  static uint32_t bmask(int i) { return ~((1u << i) - 1); }
  void get_sig(int i) {
    sigst[0] = msb[i];
    for (int d = 1; d <= 8; d++) {
      sigst[2 * d - 1] = i > d - 1 ? msb[i - d] : 0;
      sigst[2 * d] = i < numsamples - d ? msb[i + d] : 0;
    }
  }
  uint32_t avg_sum(int n) {
    uint64_t nsum = 0; int nidx = 0;
    for (int k = sample - n; k <= sample + n; k++)
      if (k >= 0 && k < numsamples) {
        int val = pabuf[k];
        val &= k < sample ? bmask(bpn) : bmask(bpn + 1);
        nsum += val; nidx++;
      }
    return nidx > 0 ? (uint32_t)((nsum + (nidx - 1)) / nidx) : 0;
  }
  static int laplace(uint32_t avg, int bpn) {
    double p_l = 0.0;
    if (avg > 0) { double theta = std::exp(-1.0 / avg); p_l = 1.0 - 1.0 / (1 + std::pow(theta, (double)(1 << bpn))); }
    return std::min(std::max((int)std::round(p_l * PSCALE), 1), (int)PSCALEm);
  }
  int predict_ref() {
    int val = pabuf[sample];
    int lval = sample > 0 ? pabuf[sample - 1] : 0, lval2 = sample > 1 ? pabuf[sample - 2] : 0;
    int nval = sample < (numsamples - 1) ? pabuf[sample + 1] : 0, nval2 = sample < (numsamples - 2) ? pabuf[sample + 2] : 0;
    int b0 = (val >> (bpn + 1)), b1 = (lval >> bpn), b2 = (nval >> (bpn + 1)), b3 = (lval2 >> bpn), b4 = (nval2 >> (bpn + 1));
    int c0 = (b0 << 1) < b1 ? 1 : 0, c1 = b0 < b2 ? 1 : 0, c2 = (b0 << 1) < b3 ? 1 : 0, c3 = b0 < b4 ? 1 : 0;
    int x0 = b0 << 1, x1 = b1, x2 = b2 << 1, x3 = b3, x4 = b4 << 1;
    int xm = (x0 + x1 + x2 + x3 + x4) / 5;
    int d0 = x0 > xm, d1 = x1 > xm;
    int ctx1 = (b0 & 15) + ((b1 & 15) << 4) + ((b2 & 15) << 8);
    int ctx2 = (c0 + (c1 << 1) + (c2 << 2) + (c3 << 3)) + (d0 << 4) + (d1 << 5);
    int ctx3 = sigst[1] + sigst[2] + sigst[3] + sigst[4] + sigst[5] + sigst[6] + sigst[7] + sigst[8];
    pl = &p_laplace[bpn]; pc1 = &cref0[msb[sample]]; pc2 = &cref1[ctx1 & 255]; pc3 = &cref2[ctx2]; pc4 = &cref3[ctx3];
    int pctx = ((((pestimate >> 12) << 1) + d0) << 1) + (b0 & 1);
    plmix = &lmixref[pctx];
    int p[5] = {pestimate, pl->p1, pc1->p1, pc2->p1, pc3->p1};
    return plmix->predict(p);
  }
  void update_ref(int bit) {
    pl->update(bit, 150); pc1->update(bit, 150); pc2->update(bit, 150); pc3->update(bit, 150); pc4->update(bit, 150);
    plmix->update(bit, 800);
    state = (state << 1) + 0;
  }
  int predict_sig() {
    int ctx1 = 0;
    for (int i = 0; i < 16; i++) if (sigst[i + 1]) ctx1 += 1 << i;
    int n1 = 0, n2 = 0;
    for (int i = 1; i <= 32; i++) {
      if (sample - i >= 0) { if (msb[sample - i]) n1++; if (msb[sample - i] > bpn) n2++; }
      if (sample + i < numsamples - 1) { if (msb[sample + i]) n1++; if (msb[sample + i] > bpn) n2++; }
    }
    pl = &p_laplace[bpn]; pc1 = &csig0[ctx1]; pc2 = &csig1[n2];
    int mixctx = ((state & 15) << 3) + ((n1 >= 3 ? 3 : n1) << 1) + (n2 > 0 ? 1 : 0);
    plmix = &lmixsig[mixctx];
    int p[3] = {pl->p1, pc1->p1, pc2->p1};
    return plmix->predict(p);
  }
  void update_sig(int bit) {
    pl->update(bit, 150); pc1->update(bit, 300); pc2->update(bit, 300);
    plmix->update(bit, 700);
    state = (state << 1) + 1;
  }
  int predict_sse(int p1) {
    int ctx1 = ((pestimate >> 11) << 1) + (sigst[0] ? 1 : 0);
    int ctx2 = 32 + (sigst[0] ? 1 : 0) + ((sigst[1] ? 1 : 0) << 1) + ((sigst[2] ? 1 : 0) << 2) + ((sigst[3] ? 1 : 0) << 3) + ((sigst[4] ? 1 : 0) << 4) + ((sigst[5] ? 1 : 0) << 5) + ((sigst[6] ? 1 : 0) << 6);
    ps1 = &sse[ctx1]; ps2 = &sse[ctx2];
    int pr1 = ps1->predict(p1), pr2 = ps2->predict(pr1);
    int p[2] = {(pr1 + pr2 + 1) >> 1, p1};
    return ssemix.predict(p);
  }
  void update_sse(int bit) { ps1->update(bit, 250); ps2->update(bit, 250); ssemix.update(bit, 250); }

  template <class F> void encode(F &&emit, int32_t *abuf) {
    pabuf = abuf;
    for (bpn = maxbpn; bpn >= 0; bpn--) {
      state = 0;
      for (sample = 0; sample < numsamples; sample++) {
        pestimate = laplace(avg_sum(32), bpn);
        get_sig(sample);
        int bit = (pabuf[sample] >> bpn) & 1;
        if (sigst[0]) { int p = predict_sse(predict_ref()); emit(p, bit); update_ref(bit); update_sse(bit); }
        else { int p = predict_sse(predict_sig()); emit(p, bit); update_sig(bit); update_sse(bit); if (bit) msb[sample] = bpn; }
      }
    }
  }
  template <class F> void decode(F &&take, int32_t *buf) {
    pabuf = buf;
    for (int i = 0; i < numsamples; i++) buf[i] = 0;
    for (bpn = maxbpn; bpn >= 0; bpn--) {
      state = 0;
      for (sample = 0; sample < numsamples; sample++) {
        pestimate = laplace(avg_sum(32), bpn);
        get_sig(sample);
        if (sigst[0]) { int bit = take(predict_sse(predict_ref())); update_ref(bit); update_sse(bit); if (bit) buf[sample] += (1 << bpn); }
        else { int bit = take(predict_sse(predict_sig())); update_sig(bit); update_sse(bit); if (bit) { buf[sample] += (1 << bpn); msb[sample] = bpn; } }
      }
    }
    for (int i = 0; i < numsamples; i++) buf[i] = u2s(buf[i]);
  }
};

// =====================================================================================
// sparse-PCM remap (libsac/map.cpp:104-202) and MapEncoder (map.cpp:3-101)
// =====================================================================================
struct Remap {
  static constexpr int scale = 1 << 15;
  std::vector<uint8_t> usedl, usedh;
  std::vector<int32_t> prefix;        // prefix[v+scale+1] = #used in [-scale, v]
  Remap() : usedl(scale + 1, 0), usedh(scale + 1, 0) {}
  void reset() { std::fill(usedl.begin(), usedl.end(), 0); std::fill(usedh.begin(), usedh.end(), 0); }
  void analyse(const int32_t *src, int n) {
    for (int i = 0; i < n; i++) {
      int v = src[i];
      if (v > 0) { if (v <= scale) usedh[v] = 1; }
      else if (v < 0) { v = -v; if (v <= scale) usedl[v] = 1; }
    }
    build();
  }
  bool is_used(int v) const {
    if (v > scale || v < -scale) return false;
    if (v > 0) return usedh[v];
    if (v < 0) return usedl[-v];
    return true;
  }
  void build() {
    prefix.assign(2 * scale + 3, 0);
    for (int v = -scale; v <= scale; v++) prefix[v + scale + 1] = prefix[v + scale] + (is_used(v) ? 1 : 0);
    prefix[2 * scale + 2] = prefix[2 * scale + 1];
  }
  int cnt(int a, int b) const {       // #used in [a,b], clipped to [-scale,scale]
    a = std::max(a, -scale); b = std::min(b, scale);
    if (a > b) return 0;
    return prefix[b + scale + 1] - prefix[a + scale];
  }
  int32_t map(int32_t pred, int32_t err) const {   // == O(|err|) loop of map.cpp:175-187
    if (err == 0) return 0;
    if (err > 0) return cnt(pred + 1, pred + err);
    return -cnt(pred + err, pred - 1);
  }
  int32_t unmap(int32_t pred, int32_t merr) const {
    if (merr == 0) return 0;
    int sgn = 1;
    if (merr < 0) { merr = -merr; sgn = -1; }
    int err = 1, terr = 0;
    while (1) { if (is_used(pred + sgn * err)) terr++; if (terr == merr) break; err++; if (err > 4 * scale) break; }
    return sgn * err;
  }
};

struct MapCoder {
  Cnt16 cnt[24], cctx[256], *pc1, *pc2, *pc3, *pc4, *px;
  Mixer mixl[4], mixh[4], finalmix, *mix;
  SSENL<32> sse0;
  std::vector<uint8_t> &ul, &uh;
  MapCoder(std::vector<uint8_t> &l, std::vector<uint8_t> &h) : ul(l), uh(h) {
    for (auto &m : mixl) m.n = 5;
    for (auto &m : mixh) m.n = 5;
    finalmix.n = 2;
  }
  int predict_low(int i) {
    int ctx1 = ul[i - 1], ctx2 = uh[i - 1], ctx3 = i > 1 ? ul[i - 2] : 0;
    pc1 = &cnt[ctx1]; pc2 = &cnt[2 + ctx2]; pc3 = &cnt[4 + (ctx1 << 1) + ctx3]; pc4 = &cnt[8 + (ctx1 << 1) + ctx2];
    int sctx = ul[i - 1];
    if (i > 1) sctx += (ul[i - 2] << 1);
    if (i > 2) sctx += (ul[i - 3] << 2);
    if (i > 3) sctx += (ul[i - 4] << 3);
    px = &cctx[sctx]; mix = &mixl[ctx1 + (ctx3 << 1)];
    int p[5] = {pc1->p1, pc2->p1, pc3->p1, pc4->p1, px->p1};
    return mix->predict(p);
  }
  int predict_high(int i) {
    int ctx1 = uh[i - 1], ctx2 = ul[i], ctx3 = i > 1 ? uh[i - 2] : 0;
    pc1 = &cnt[12 + ctx1]; pc2 = &cnt[12 + 2 + ctx2]; pc3 = &cnt[12 + 4 + (ctx1 << 1) + ctx3]; pc4 = &cnt[12 + 8 + (ctx1 << 1) + ctx2];
    int sctx = uh[i - 1];
    if (i > 1) sctx += (uh[i - 2] << 1);
    if (i > 2) sctx += (uh[i - 3] << 2);
    if (i > 3) sctx += (uh[i - 4] << 3);
    px = &cctx[32 + sctx]; mix = &mixh[ctx1 + (ctx3 << 1)];
    int p[5] = {pc1->p1, pc2->p1, pc3->p1, pc4->p1, px->p1};
    return mix->predict(p);
  }
  void update(int bit) { pc1->update(bit, 500); pc2->update(bit, 500); pc3->update(bit, 500); pc4->update(bit, 500); px->update(bit, 500); mix->update(bit, 1000); }
  int predict_sse(int p1) { int p[2] = {sse0.predict(p1), p1}; return finalmix.predict(p); }
  void update_sse(int bit) { sse0.update(bit, 300); finalmix.update(bit, 500); }
  void encode(RangeEnc &rc) {
    for (int i = 1; i <= 1 << 15; i++) {
      int bit = ul[i];
      rc.encode(predict_sse(predict_low(i)), bit); update(bit); update_sse(bit);
      bit = uh[i];
      rc.encode(predict_sse(predict_high(i)), bit); update(bit); update_sse(bit);
    }
  }
  void decode(RangeDec &rc) {
    for (int i = 1; i <= 1 << 15; i++) {
      int bit = rc.decode(predict_sse(predict_low(i))); update(bit); ul[i] = (uint8_t)bit; update_sse(bit);
      bit = rc.decode(predict_sse(predict_high(i))); update(bit); uh[i] = (uint8_t)bit; update_sse(bit);
    }
  }
};

// =====================================================================================
// cost functions (libsac/cost.h)
// =====================================================================================
static double cost_l1(const int32_t *b, int n) {
  if (!n) return 0.;
  int64_t sum = 0;
  for (int i = 0; i < n; i++) sum = (int64_t)((double)sum + std::fabs((double)b[i]));
  return sum / static_cast<double>(n);
}
static double cost_rms(const int32_t *b, int n) {
  if (!n) return 0.;
  int64_t sum = 0;
  for (int i = 0; i < n; i++) sum += (int32_t)((uint32_t)b[i] * (uint32_t)b[i]);
  return std::sqrt(sum / static_cast<double>(n));
}
static double cost_golomb(const int32_t *b, int n) {
  if (!n) return 0;
  double rm = 0.0; int64_t nbits = 0;
  for (int i = 0; i < n; i++) {
    const int32_t m = std::max(static_cast<int32_t>(rm), 1);
    const int32_t uval = s2u(b[i]);
    int q = uval / m;
    nbits += (q + 1);
    if (m > 1) nbits += (32 - __builtin_clz((uint32_t)m));
    rm = std::fma(0.97, rm, (double)uval);
  }
  return nbits / 8.;
}
static double cost_entropy(const int32_t *b, int n) {
  double entropy = 0.0;
  if (!n) return entropy;
  int32_t mn = std::numeric_limits<int32_t>::max(), mx = std::numeric_limits<int32_t>::min();
  for (int i = 0; i < n; i++) { mx = std::max(mx, b[i]); mn = std::min(mn, b[i]); }
  std::vector<int> counts((size_t)mx - mn + 1, 0);
  for (int i = 0; i < n; i++) ++counts[b[i] - mn];
  const double invs = 1.0 / static_cast<double>(n);
  if (counts.size() < (size_t)n) {
    for (int c : counts) { if (c == 0) continue; const double p = c * invs; entropy = std::fma((double)c, std::log2(p), entropy); }
  } else {
    for (int i = 0; i < n; i++) { const double p = counts[b[i] - mn] * invs; entropy += std::log2(p); }
  }
  return -entropy / 8.0;
}
static double cost_bitplane(const int32_t *b, int n) {
  std::vector<int32_t> u(n); int vmax = 0;
  for (int i = 0; i < n; i++) { int v = s2u(b[i]); if (v > vmax) vmax = v; u[i] = v; }
  RangeEnc rc; Bitplane bc(ilog2(vmax), n);
  bc.encode([&](int p, int bit) { rc.encode(p, bit); }, u.data());
  rc.stop();
  return (double)rc.out.size();
}
static double cost(int kind, const int32_t *b, int n) {
  switch (kind) { case 0: return cost_l1(b, n); case 1: return cost_rms(b, n); case 2: return cost_entropy(b, n); case 3: return cost_golomb(b, n); case 4: return cost_bitplane(b, n); }
  return -1;
}

// =====================================================================================
// DDS search (opt/opt.cpp, opt/dds.cpp, opt/ssc.h, common/rand.h)
// =====================================================================================
struct Box { double xmin, xmax; };
struct DDS {
  std::mt19937 eng{0};                                         // opt.cpp:5
  std::vector<Box> pb; int ndim; int nfunc_max, num_threads; double sigma_init;
  int c_succ_max = 3, c_fail_max = 50;
  double r01() { return std::uniform_real_distribution<double>{0, 1}(eng); }
  // std::normal_distribution<double>{0,1} freshly constructed per draw (common/rand.h:25-27):
  // libstdc++'s Marsaglia polar method (bits/random.tcc), returning y*mult; written out because
  // the reference build (-ffp-contract=fast) contracts its x*x+y*y into fma(x,x,y*y).
  double rnorm() {
    double x, y, r2;
    do {
      x = 2.0 * r01() - 1.0;
      y = 2.0 * r01() - 1.0;
      r2 = std::fma(x, x, y * y);
    } while (r2 > 1.0 || r2 == 0.0);
    const double mult = std::sqrt(-2 * std::log(r2) / r2);
    return y * mult;
  }
  uint32_t ruint(uint32_t a, uint32_t b) { return std::uniform_int_distribution<uint32_t>{a, b}(eng); }
  static double reflect(double x, double lo, double hi) {
    if (x < lo) { x = lo + (lo - x); if (x > hi) x = lo; }
    if (x > hi) { x = hi - (x - hi); if (x < lo) x = hi; }
    return x;
  }
  double gen_norm(double x, const Box &b, double r) {
    double sigma = r * (b.xmax - b.xmin);
    double xn = std::fma(sigma, rnorm(), x);
    return reflect(xn, b.xmin, b.xmax);
  }
  std::vector<double> candidate(const std::vector<double> &x, int nfunc, double sigma) {
    std::vector<int> J;
    double p = 1.0 - std::log((double)nfunc) / std::log((double)nfunc_max);
    for (int i = 0; i < ndim; i++) if (r01() < p) J.push_back(i);
    if (J.empty()) J.push_back((int)ruint(0, ndim - 1));
    std::vector<double> xt = x;
    for (int k : J) xt[k] = gen_norm(x[k], pb[k], sigma);
    return xt;
  }
  template <class F> std::pair<double, std::vector<double>> run(F &&f, const std::vector<double> &xs) {
    std::pair<double, std::vector<double>> xb{f(xs), xs};
    double sigma = sigma_init; int nfunc = 1;
    if (num_threads <= 0) {
      int nsucc = 0, nfail = 0;                                 // SSC0(3,50)
      while (nfunc < nfunc_max) {
        auto xg = candidate(xb.second, nfunc, sigma);
        double c = f(xg); nfunc++;
        double lam = 0.0;
        if (c < xb.first) { xb = {c, xg}; lam = 1.0; }
        if (lam > 0.0) { nsucc++; nfail = 0; } else { nsucc = 0; nfail++; }
        if (nsucc >= c_succ_max) { sigma *= 2.0; nsucc = 0; } else if (nfail >= c_fail_max) { sigma /= 2.0; nfail = 0; }
        sigma = std::min(std::max(sigma, 0.05), 0.5);
      }
    } else {
      double p_succ = 0.05; const double p_t = 0.05, p_c = 0.10, p_d = 0.05;   // SSC1(0.05,0.10,0.05)
      while (nfunc < nfunc_max) {
        const int nt = std::min(nfunc_max - nfunc, num_threads);
        std::vector<std::pair<double, std::vector<double>>> g(nt);
        for (int i = 0; i < nt; i++) { g[i].second = candidate(xb.second, nfunc, sigma); nfunc++; }
        for (int i = 0; i < nt; i++) g[i].first = f(g[i].second);
        const double old = xb.first; int ns = 0;
        for (auto &x : g) if (x.first < old) { ns++; if (x.first < xb.first) xb = x; }
        const double lam = ns / static_cast<double>(nt);
        p_succ = std::fma(1.0 - p_c, p_succ, p_c * lam);
        sigma = sigma * std::exp(p_d * (p_succ - p_t) / (1.0 - p_t));
        sigma = std::min(std::max(sigma, 0.05), 0.25);
      }
    }
    return xb;
  }
};

// =====================================================================================
// The two other searchers FrameCoder::Optimize can be given (libsac.cpp:408-415): OptDE (opt/de.cpp:10-184, opt/de.h) and
// OptCMA (opt/cma.cpp:6-92, opt/cma.h).  Both reuse the draws, gen_norm and reflect of the class above (Opt base, opt/opt.cpp);
// fused multiply-adds as the reference build contracts them (checked bit for bit against oracle/_ref: tests/golden r4).
// =====================================================================================
struct DE : DDS {
  // DECfg defaults (de.h:19-31): NP 30, CR 0.5, F 0.5, c 0.1, CURPBEST, INIT_NORM, pbest 0.1 -> npbest = clamp(round(3) - 1, 0, 29) = 2
  static constexpr int NP = 30, NPBEST = 2;
  using Point = std::pair<double, std::vector<double>>;
  double gen_CR(double mCR) { return std::min(std::max(std::fma(rnorm(), 0.1, mCR), 0.01), 1.0); }       // normal_distribution{mCR, 0.1}: ret * sd + mean, fused
  double gen_F(double mF) {                                                                              // cauchy_distribution{mF, 0.1} (bits/random.tcc): a + b * tan(pi * u), unfused
    double u;
    do u = r01(); while (u == 0.5);
    const double pi = 3.1415926535897932384626433832795029L;
    return std::min(std::max(mF + 0.1 * std::tan(pi * u), 0.01), 1.0);
  }
  std::vector<int> select_k_unique_except(int n, int ie, int k) {                                        // de.cpp:12-29
    std::vector<int> r;
    if (k >= n - 1) return r;
    std::vector<int> e;
    for (int i = 0; i < n; i++) if (i != ie) e.push_back(i);
    for (int i = 0; i < k; i++) { const int idx = (int)ruint(0, (uint32_t)e.size() - 1); r.push_back(e[idx]); e.erase(e.begin() + idx); }
    return r;
  }
  template <class F> Point run(F &&f, const std::vector<double> &xs) {                                   // de.cpp:77-164
    int nfunc = 1;
    Point xb{f(xs), xs};
    std::vector<Point> pop(NP);
    pop[0] = xb;
    for (int a = 1; a < NP; a++) {                                                                       // gen_norm_samples (opt.cpp:125-133)
      std::vector<double> xt(ndim);
      for (int i = 0; i < ndim; i++) xt[i] = gen_norm(xb.second[i], pb[i], sigma_init);
      pop[a].second = xt;
    }
    for (int a = 1; a < NP; a++) { pop[a].first = f(pop[a].second); nfunc++; }                            // the whole start-up population, whatever nfunc_max
    for (int a = 1; a < NP; a++) if (pop[a].first < xb.first) xb = pop[a];
    double mCR = 0.5, mF = 0.5;
    while (nfunc < nfunc_max) {
      std::sort(pop.begin(), pop.end(), [](const Point &a, const Point &b) { return a.first < b.first; });
      const int agents = std::min(nfunc_max - nfunc, (int)pop.size());
      std::vector<Point> gen(agents);
      std::vector<std::pair<double, double>> mut(agents);
      for (int ia = 0; ia < agents; ia++) {                                                              // generate_candidate, de.cpp:31-70
        const double tCR = gen_CR(mCR), tF = gen_F(mF);
        const int R = (int)ruint(0, ndim - 1);
        const std::vector<int> v = select_k_unique_except((int)pop.size(), ia, 2);
        const int np = std::min(NPBEST, (int)pop.size() - 1);
        const int xp = np > 0 ? (int)ruint(0, np) : 0;
        const std::vector<double> &pbest = pop[xp].second, &cur = pop[ia].second, &x1 = pop[v[0]].second, &x2 = pop[v[1]].second;
        std::vector<double> xm(ndim), xt(ndim);
        for (int i = 0; i < ndim; i++)                                                                   // mut_curbest, de.cpp:176-184
          xm[i] = reflect(std::fma(tF, x1[i] - x2[i], std::fma(tF, pbest[i] - cur[i], cur[i])), pb[i].xmin, pb[i].xmax);
        for (int i = 0; i < ndim; i++) xt[i] = (r01() < tCR || i == R) ? xm[i] : cur[i];
        gen[ia].second = xt; mut[ia] = {tCR, tF};
      }
      for (int ia = 0; ia < agents; ia++) { gen[ia].first = f(gen[ia].second); nfunc++; }
      std::vector<double> crs, fs;
      for (int ia = 0; ia < agents; ia++)
        if (gen[ia].first < pop[ia].first) {
          pop[ia] = gen[ia]; crs.push_back(mut[ia].first); fs.push_back(mut[ia].second);
          if (pop[ia].first < xb.first) xb = pop[ia];
        }
      if (nfunc >= nfunc_max) break;
      double mean = 0.0, lehmer = 0.0;                                                                   // MathUtils::mean / meanL, utils.h:283-305
      if (!crs.empty()) { double sum = 0.0; for (double v2 : crs) sum += v2; mean = sum / static_cast<double>(crs.size()); }
      if (!fs.empty()) {
        double s0 = 0.0, s1 = 0.0; size_t k = 0;
        for (; k + 4 <= fs.size(); k += 4) for (size_t q = k; q < k + 4; q++) { s0 = s0 + fs[q] * fs[q]; s1 += fs[q]; }   // vectorised squares, in-order adds
        for (; k < fs.size(); k++) { s0 = std::fma(fs[k], fs[k], s0); s1 += fs[k]; }
        if (s1 > 0.0) lehmer = s0 / s1;
      }
      mCR = std::fma(mean, 0.1, (1.0 - 0.1) * mCR);
      mF = std::fma(lehmer, 0.1, (1.0 - 0.1) * mF);
    }
    return xb;
  }
};

struct CMA : DDS {
  template <class F> std::pair<double, std::vector<double>> run(F &&f, const std::vector<double> &xs) {  // cma.cpp:49-92
    const int n = ndim;
    const double d = 1.0 + n / 2.0, p_t = 2.0 / 11.0, cp = 1.0 / 12.0, cc = 2.0 / (n + 2.0), ccov = 2.0 / (n * n + 6.0);   // CMAParams, cma.h:17-38
    std::vector<double> pc(n, 0.0), mcov((size_t)n * n, 0.0), G((size_t)n * n, 0.0), az(n), z(n), xg(n);
    for (int i = 0; i < n; i++) mcov[(size_t)i * n + i] = 1.0;
    double sigma = sigma_init, p_succ = p_t;                    // SSC1(p_target, cp, 1/d), ssc.h:43-60 (bounds 0.05 .. 0.25)
    std::pair<double, std::vector<double>> xb{f(xs), xs};
    int nfunc = 1;
    auto fold_sub = [](double acc, int m, const double *a, const double *b) {   // in-order reduction as the vectorised loop runs it
      int k = 0;
      for (; k + 4 <= m; k += 4) { acc = acc - a[k] * b[k]; acc = acc - a[k + 1] * b[k + 1]; acc = acc - a[k + 2] * b[k + 2]; acc = acc - a[k + 3] * b[k + 3]; }
      if (m - k >= 2) { acc = acc - a[k] * b[k]; acc = acc - a[k + 1] * b[k + 1]; k += 2; }
      if (k < m) acc = std::fma(-a[k], b[k], acc);
      return acc;
    };
    while (nfunc < nfunc_max) {
      // slmath::Cholesky::Factor(mcov, 0.1) (math.h:89-111); a failure leaves G half-updated and is ignored, as in the reference
      for (int i = 0; i < n; i++) std::copy_n(&mcov[(size_t)i * n], i + 1, &G[(size_t)i * n]);
      for (int i = 0; i < n; i++) {
        double *gi = &G[(size_t)i * n];
        for (int j = 0; j < i; j++) { const double *gj = &G[(size_t)j * n]; gi[j] = fold_sub(gi[j], j, gi, gj) / gj[j]; }
        const double s = fold_sub(gi[i] + 0.1, i, gi, gi);
        if (s > 1E-8) gi[i] = std::sqrt(s); else break;
      }
      for (int i = 0; i < n; i++) z[i] = rnorm();
      for (int i = 0; i < n; i++) az[i] = dot_ref(&G[(size_t)i * n], z.data(), (size_t)n);                         // slmath::mul(G, z)
      for (int i = 0; i < n; i++) xg[i] = reflect(std::fma((pb[i].xmax - pb[i].xmin) * sigma, az[i], xb.second[i]), pb[i].xmin, pb[i].xmax);
      const double fn = f(xg);
      const double lam = fn < xb.first ? 1.0 : 0.0;
      p_succ = std::fma(1.0 - cp, p_succ, cp * lam);
      sigma = sigma * std::exp((1.0 / d) * (p_succ - p_t) / (1.0 - p_t));
      sigma = std::min(std::max(sigma, 0.05), 0.25);
      if (fn < xb.first) {
        xb = {fn, xg};
        const double a = 1.0 - cc, b = std::sqrt(cc * (2.0 - cc));                                        // update_cov, cma.cpp:45-49
        for (int i = 0; i < n; i++) pc[i] = std::fma(a, pc[i], b * az[i]);
        for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) mcov[(size_t)j * n + i] = std::fma(1.0 - ccov, mcov[(size_t)j * n + i], ccov * (pc[j] * pc[i]));
      }
      nfunc++;
    }
    return xb;
  }
};
static int g_orc_search = 0;     // FrameCoder::SearchMethod of the next orc_encode_frame calls: 0 DDS, 1 DE, 2 CMA

// =====================================================================================
// frame statistics, encode and decode (libsac.cpp:201-298,429-479,507-593,626-651)
// =====================================================================================
static void analyse(const int32_t *src, int n, int32_t *mean, int32_t *mn, int32_t *mx) {
  int64_t sum = 0;
  for (int i = 0; i < n; i++) sum += src[i];
  *mean = (int)std::floor(sum / (double)n);
  int32_t lo = std::numeric_limits<int32_t>::max(), hi = std::numeric_limits<int32_t>::min();
  for (int i = 0; i < n; i++) { hi = std::max(hi, src[i]); lo = std::min(lo, src[i]); }
  *mn = lo; *mx = hi;
}
static void put32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
static void put16(std::vector<uint8_t> &o, uint32_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }
static uint32_t get32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

static int encode_frame(int nch, int framesize, int n, const int32_t *raw, const orc_frame_cfg &rc,
                        float *profile_io, std::vector<uint8_t> &rec, double *trace_cost,
                        float *trace_coefs, int *info) {
  Profile base; base.load_default();
  if (profile_io) for (int i = 0; i < 58; i++) base.c[i].vdef = profile_io[i];
  std::vector<int32_t> smp((size_t)nch * n);
  std::vector<int32_t> stats(3 * nch);
  std::vector<Remap> maps(nch);
  int32_t mean[2] = {0, 0};
  for (int ch = 0; ch < nch; ch++) {
    int32_t m, lo, hi;
    analyse(raw + (size_t)ch * n, n, &m, &lo, &hi);
    if (rc.sparse_pcm) { maps[ch].reset(); maps[ch].analyse(raw + (size_t)ch * n, n); }
    if (rc.zero_mean == 0) m = 0;
    for (int i = 0; i < n; i++) smp[(size_t)ch * n + i] = raw[(size_t)ch * n + i] - m;
    lo -= m; hi -= m;
    mean[ch] = m; stats[3 * ch] = lo; stats[3 * ch + 1] = hi; stats[3 * ch + 2] = m;
  }
  float coefs[58];
  if (rc.optimize) {
    if (rc.reset) base.load_default();
    std::vector<int> lp;
    for (int i = 0; i < 58; i++) if (i != 56 && i != 57) lp.push_back(i);
    const int ndim = (int)lp.size();
    int nopt = std::min(n, static_cast<int>(std::ceil(framesize * rc.fraction)));
    const int start = (n - nopt) / 2;
    DDS dds; dds.ndim = ndim; dds.pb.resize(ndim); dds.nfunc_max = rc.maxnfunc; dds.num_threads = rc.num_threads; dds.sigma_init = rc.sigma;
    std::vector<double> xs(ndim);
    for (int i = 0; i < ndim; i++) { dds.pb[i] = {base.c[lp[i]].vmin, base.c[lp[i]].vmax}; xs[i] = base.c[lp[i]].vdef; }
    int neval = 0;
    std::vector<int32_t> err((size_t)nch * nopt);
    auto f = [&](const std::vector<double> &x) {
      float g[58];
      for (int i = 0; i < 58; i++) g[i] = base.c[i].vdef;
      for (int i = 0; i < ndim; i++) g[lp[i]] = (float)x[i];
      predict_frame(nch, n, smp.data(), stats.data(), g, start, nopt, true, rc.optk, err.data(), nullptr, nullptr);
      double c = 0.0;
      for (int ch = 0; ch < nch; ch++) c += cost(rc.cost, err.data() + (size_t)ch * nopt, nopt);
      if (neval < rc.maxnfunc + (g_orc_search ? 32 : 0)) {      // DE evaluates its whole start-up population (30 points) even when maxnfunc is smaller
        if (trace_cost) trace_cost[neval] = c;
        if (trace_coefs) std::memcpy(trace_coefs + (size_t)neval * 58, g, sizeof(g));
      }
      neval++;
      return c;
    };
    std::pair<double, std::vector<double>> best;
    if (g_orc_search == 1) { DE de; static_cast<DDS &>(de) = dds; best = de.run(f, xs); }           // cmdline.cpp:221-241: same maxnfunc / sigma
    else if (g_orc_search == 2) { CMA cma; static_cast<DDS &>(cma) = dds; best = cma.run(f, xs); }
    else best = dds.run(f, xs);
    for (int i = 0; i < ndim; i++) base.c[lp[i]].vdef = (float)best.second[i];
  }
  for (int i = 0; i < 58; i++) coefs[i] = base.c[i].vdef;
  std::vector<int32_t> error((size_t)nch * n), pred((size_t)nch * n);
  predict_frame(nch, n, smp.data(), stats.data(), coefs, 0, n, false, rc.optk, error.data(), pred.data(), nullptr);

  rec.clear();
  put32(rec, (uint32_t)n);
  for (int i = 0; i < 58; i++) { uint32_t ix; std::memcpy(&ix, &coefs[i], 4); put32(rec, ix); }
  for (int ch = 0; ch < nch; ch++) {
    const int32_t *e = error.data() + (size_t)ch * n;
    std::vector<int32_t> u(n); int32_t emax = 0;
    for (int i = 0; i < n; i++) { u[i] = s2u(e[i]); emax = std::max(emax, u[i]); }
    int maxbpn = ilog2(emax);
    std::vector<uint8_t> enc;
    bool mapped = false; int maxbpn_map = 0;
    {
      RangeEnc r; Bitplane bc(maxbpn, n);
      std::vector<int32_t> tmp = u;
      bc.encode([&](int p, int bit) { r.encode(p, bit); }, tmp.data());
      r.stop(); enc = r.out;
    }
    if (rc.sparse_pcm) {
      std::vector<int32_t> um(n), em(n); int32_t emx = 0;
      for (int i = 0; i < n; i++) { em[i] = maps[ch].map(pred[(size_t)ch * n + i], e[i]); um[i] = s2u(em[i]); emx = std::max(emx, um[i]); }
      maxbpn_map = ilog2(emx);
      double ent1 = cost_l1(e, n), ent2 = cost_l1(em.data(), n), r = 1.0;
      if (ent2 != 0.0) r = ent1 / ent2;
      if (r > 1.05) {
        RangeEnc rr; MapCoder me(maps[ch].usedl, maps[ch].usedh); me.encode(rr);
        Bitplane bc(maxbpn_map, n);
        bc.encode([&](int p, int bit) { rr.encode(p, bit); }, um.data());
        rr.stop();
        if (rr.out.size() < enc.size()) { mapped = true; enc = rr.out; }
      }
    }
    put32(rec, (uint32_t)enc.size()); put32(rec, (uint32_t)mean[ch]); put32(rec, (uint32_t)stats[3 * ch]); put32(rec, (uint32_t)stats[3 * ch + 1]);
    uint32_t flag = mapped ? ((1u << 9) | (uint32_t)maxbpn_map) : (uint32_t)maxbpn;
    put16(rec, flag);
    rec.insert(rec.end(), enc.begin(), enc.end());
    if (info) { info[3 * ch] = mapped ? maxbpn_map : maxbpn; info[3 * ch + 1] = mapped; info[3 * ch + 2] = (int)enc.size(); }
  }
  if (profile_io) for (int i = 0; i < 58; i++) profile_io[i] = coefs[i];
  return (int)rec.size();
}

// decoder: ReadEncoded + Decode + Unpredict (libsac.cpp:144-199,280-298,580-593)
static int decode_frame(const uint8_t *rec, int len, int nch, int framesize, int32_t *out, int cap, float *coefs_out) {
  if (len < 4 + 232) return -1;
  int n = (int)get32(rec);
  if (n > cap || n > framesize || n < 0) return -2;
  float coefs[58];
  for (int i = 0; i < 58; i++) { uint32_t ix = get32(rec + 4 + 4 * i); std::memcpy(&coefs[i], &ix, 4); }
  size_t pos = 4 + 232;
  std::vector<int32_t> err((size_t)nch * n);
  int32_t lo[2], hi[2], mean[2]; bool mapped[2] = {false, false};
  std::vector<Remap> maps(nch);
  for (int ch = 0; ch < nch; ch++) {
    uint32_t bs = get32(rec + pos); mean[ch] = (int32_t)get32(rec + pos + 4); lo[ch] = (int32_t)get32(rec + pos + 8); hi[ch] = (int32_t)get32(rec + pos + 12);
    uint32_t flag = rec[pos + 16] | (rec[pos + 17] << 8);
    pos += 18;
    mapped[ch] = (flag >> 9) != 0; int maxbpn = flag & 0xff;
    RangeDec rd(rec + pos, bs);
    if (mapped[ch]) { maps[ch].reset(); MapCoder me(maps[ch].usedl, maps[ch].usedh); me.decode(rd); }
    Bitplane bc(maxbpn, n);
    bc.decode([&](int p) { return rd.decode(p); }, err.data() + (size_t)ch * n);
    pos += bs;
  }
  if (nch == 1) { lo[1] = lo[0]; hi[1] = hi[0]; mean[1] = mean[0]; }
  Params P; set_param(P, coefs, false, 4);
  Predictor pr; pr.init(P, lo, hi);
  auto step = [&](int chp, int ch, int32_t *dst, int idx) {
    const double pd = pr.c[chp].predict();
    const int32_t pi = std::min(std::max(cvt_i32_x86(std::round(pd)), lo[ch]), hi[ch]);
    const int32_t e = err[(size_t)ch * n + idx];
    dst[idx] = mapped[ch] ? pi + maps[ch].unmap(pi + mean[ch], e) : pi + e;
    pr.c[chp].update(dst[idx]);
  };
  if (nch == 1) {
    for (int idx = 0; idx < n; idx++) { pr.fill0(out, idx, out, idx); step(0, 0, out, idx); }
  } else {
    int ch0 = P.ch_ref, ch1 = 1 - ch0;
    int32_t *d0 = out + (size_t)ch0 * n, *d1 = out + (size_t)ch1 * n;
    int i0 = 0, i1 = 0;
    while (i0 < n || i1 < n) {
      if (i0 < n) { pr.fill0(d0, i0, d1, i1); step(0, ch0, d0, i0); i0++; }
      if (i0 >= P.nS1) { pr.fill1(d0, d1, i1, n); step(1, ch1, d1, i1); i1++; }
    }
  }
  for (int ch = 0; ch < nch; ch++) if (mean[ch] != 0) for (int i = 0; i < n; i++) out[(size_t)ch * n + i] += mean[ch];
  if (coefs_out) std::memcpy(coefs_out, coefs, sizeof(coefs));
  return n;
}

} // namespace orc

// =====================================================================================
// C ABI
// =====================================================================================
using namespace orc;

API int orc_profile(float *out) {
  Profile p; p.load_default();
  for (int i = 0; i < 58; i++) { out[3 * i] = p.c[i].vmin; out[3 * i + 1] = p.c[i].vmax; out[3 * i + 2] = p.c[i].vdef; }
  return 58;
}
API void orc_domain_tables(int *fwd, int *inv) {
  for (int i = 0; i < PSCALE; i++) fwd[i] = dom().Fwd(i);
  for (int x = -2047; x <= 2047; x++) inv[x + 2047] = dom().Inv(x);
}
API int orc_predict_frame(int nch, int, int total, const int32_t *samples, const int32_t *stats, const float *coefs, int from, int n, int optimize, int optk, int32_t *error, int32_t *pred) {
  predict_frame(nch, total, samples, stats, coefs, from, n, optimize != 0, optk, error, (pred && !optimize) ? pred : nullptr, nullptr);
  return 0;
}
API int orc_predict_trace(int nch, int total, const int32_t *samples, const int32_t *stats, const float *coefs, int from, int n, int optimize, int optk, double *pd, double *plpc, double *plms, int32_t *error) {
  Trace t{pd, plpc, plms};
  predict_frame(nch, total, samples, stats, coefs, from, n, optimize != 0, optk, error, nullptr, &t);
  return 0;
}
API double orc_cost(int kind, const int32_t *buf, int n) { return cost(kind, buf, n); }
API int orc_bitplane_encode(const int32_t *s, int n, int maxbpn, uint8_t *out, int cap) {
  RangeEnc r; Bitplane bc(maxbpn, n); std::vector<int32_t> tmp(s, s + n);
  bc.encode([&](int p, int bit) { r.encode(p, bit); }, tmp.data());
  r.stop();
  int len = (int)r.out.size();
  if (len > cap) return -len;
  std::memcpy(out, r.out.data(), len);
  return len;
}
API int orc_bitplane_trace(const int32_t *s, int n, int maxbpn, uint16_t *p1s, uint8_t *bits, int maxdec) {
  Bitplane bc(maxbpn, n); std::vector<int32_t> tmp(s, s + n); int cnt = 0;
  bc.encode([&](int p, int bit) { if (cnt < maxdec) { p1s[cnt] = (uint16_t)p; bits[cnt] = (uint8_t)bit; } cnt++; }, tmp.data());
  return cnt;
}
API int orc_bitplane_decode(const uint8_t *in, int len, int n, int maxbpn, int32_t *err_out) {
  RangeDec rd(in, len); Bitplane bc(maxbpn, n);
  bc.decode([&](int p) { return rd.decode(p); }, err_out);
  return 0;
}
API int orc_rangecoder_encode(const uint16_t *p1s, const uint8_t *bits, int n, uint8_t *out, int cap) {
  RangeEnc r;
  for (int i = 0; i < n; i++) r.encode(p1s[i], bits[i]);
  r.stop();
  int len = (int)r.out.size();
  if (len > cap) return -len;
  std::memcpy(out, r.out.data(), len);
  return len;
}
API double orc_remap(const int32_t *raw, int n, const int32_t *pred, const int32_t *error, int32_t *s2u_map, int *maxbpn_map, uint8_t *usedl, uint8_t *usedh) {
  Remap m; m.reset(); m.analyse(raw, n);
  std::vector<int32_t> em(n); int32_t emx = 0;
  for (int i = 0; i < n; i++) { em[i] = m.map(pred[i], error[i]); s2u_map[i] = s2u(em[i]); emx = std::max(emx, s2u_map[i]); }
  *maxbpn_map = ilog2(emx);
  double e1 = cost_l1(error, n), e2 = cost_l1(em.data(), n), r = 1.0;
  if (e2 != 0.0) r = e1 / e2;
  if (usedl) std::memcpy(usedl, m.usedl.data(), m.usedl.size());
  if (usedh) std::memcpy(usedh, m.usedh.data(), m.usedh.size());
  return r;
}
API int orc_mapencode(const uint8_t *usedl, const uint8_t *usedh, uint8_t *out, int cap) {
  std::vector<uint8_t> l(usedl, usedl + (1 << 15) + 1), h(usedh, usedh + (1 << 15) + 1);
  RangeEnc r; MapCoder me(l, h); me.encode(r); r.stop();
  int len = (int)r.out.size();
  if (len > cap) return -len;
  std::memcpy(out, r.out.data(), len);
  return len;
}

// ---------------------------------------------------------------- adaptive sub-frame split
// SparsePCM::Analyse (libsac/sparse.h:31-73): fraction of the value range [min,max] in use and
// the ratio sum|val| / sum|rank(val)| where rank counts only used values between 0 and val
// (val2rank_fast, sparse.h:77-96, with p = 0).
static void sparse_cost(const int32_t *buf, int n, double *used_pct, double *cost) {
  *used_pct = 0.0; *cost = 0.0;
  if (n <= 0) return;
  int32_t mn = buf[0], mx = buf[0];
  for (int i = 0; i < n; i++) { if (buf[i] > mx) mx = buf[i]; if (buf[i] < mn) mn = buf[i]; }
  const int N = mx - mn + 1;
  std::vector<int> used(N, 0), prefix(N + 1, 0);
  for (int i = 0; i < n; i++) used[buf[i] - mn] = 1;
  for (int i = 0; i < N; i++) prefix[i + 1] = prefix[i] + used[i];
  *used_pct = (prefix[N] / static_cast<double>(N)) * 100.;
  double sum0 = 0, sum1 = 0;
  const int pidx = 0 - mn;
  for (int i = 0; i < n; i++) {
    const int32_t v = buf[i];
    int r = 0;
    if (v > 0) r = prefix[v - mn + 1] - prefix[std::min(std::max(pidx + 1, 0), N)];
    else if (v < 0) r = prefix[v - mn] - prefix[std::min(std::max(pidx, 0), N)];
    sum0 += std::fabs((double)v);
    sum1 += std::fabs((double)r);
  }
  *cost = sum1 > 0 ? sum0 / sum1 : 0;
}

// Codec::Analyse + PushState (libsac/libsac.cpp:705-780): blocks of `blocksamples`, state = mean over
// channels of the sparse cost > 1.35; runs of equal state form a sub-frame, a run shorter than
// min_frame_length is appended to the previous sub-frame (if there is one).
struct SubFrame { int state, start, length; };
static void push_state(std::vector<SubFrame> &sf, SubFrame &cur, int min_len, int block_state, int samples_block) {
  if (block_state == cur.state) cur.length += samples_block;
  else {
    if (cur.length < min_len && !sf.empty()) sf.back().length += cur.length;   // extend (cur is left as it is: libsac.cpp:711-714)
    else {
      sf.push_back(cur);
      if (samples_block) { cur.state = block_state; cur.start += cur.length; cur.length = samples_block; }
    }
  }
}
API int orc_plan_subframes(int nch, int samples_read, const int32_t *pcm, long long ch_stride, int blocksamples,
                           int min_frame_length, int *out, int cap) {
  std::vector<SubFrame> sf;
  SubFrame cur{-1, 0, 0};
  int done = 0, nblock = 0;
  while (done < samples_read) {
    const int nb = std::min(blocksamples, samples_read - done);
    double avg_cost = 0;
    for (int ch = 0; ch < nch; ch++) { double u, c; sparse_cost(pcm + (size_t)ch * ch_stride + done, nb, &u, &c); avg_cost += c; }
    avg_cost /= (double)nch;
    const int st = avg_cost > 1.35;
    if (nblock == 0) { cur.state = st; cur.length = nb; cur.start = 0; }
    else push_state(sf, cur, min_frame_length, st, nb);
    done += nb; nblock++;
  }
  if (cur.length) push_state(sf, cur, min_frame_length, -1, 0);
  int n = 0;
  for (auto &f : sf) { if (n < cap) { out[3 * n] = f.start; out[3 * n + 1] = f.length; out[3 * n + 2] = f.state; } n++; }
  return n;
}
API void orc_sparse_cost(const int32_t *buf, int n, double *out2) { sparse_cost(buf, n, &out2[0], &out2[1]); }

API void orc_analyse(const int32_t *raw, int n, int32_t *out) { analyse(raw, n, &out[0], &out[1], &out[2]); }
API void orc_rng(int n, const int *kinds, const double *args, double *out) {
  DDS d;
  for (int i = 0; i < n; i++) {
    if (kinds[i] == 0) out[i] = d.r01(); else if (kinds[i] == 1) out[i] = d.rnorm(); else out[i] = (double)d.ruint(0, (uint32_t)args[i]);
  }
}
API void orc_gen_norm(double x, double xmin, double xmax, double r, int n, double *out) {
  DDS d; Box b{xmin, xmax};
  for (int i = 0; i < n; i++) out[i] = d.gen_norm(x, b, r);
}
API void orc_set_search_method(int search) { g_orc_search = search; }
API double orc_reflect(double x, double xmin, double xmax) { return DDS::reflect(x, xmin, xmax); }
API void orc_ssc(int which, int n, const double *lam, double sigma0, double *out) {
  double sigma = sigma0; int nsucc = 0, nfail = 0; double p_succ = 0.05;
  for (int i = 0; i < n; i++) {
    if (which == 0) {
      if (lam[i] > 0.0) { nsucc++; nfail = 0; } else { nsucc = 0; nfail++; }
      if (nsucc >= 3) { sigma *= 2.0; nsucc = 0; } else if (nfail >= 50) { sigma /= 2.0; nfail = 0; }
      sigma = std::min(std::max(sigma, 0.05), 0.5);
    } else {
      p_succ = std::fma(1.0 - 0.10, p_succ, 0.10 * lam[i]);
      sigma = sigma * std::exp(0.05 * (p_succ - 0.05) / (1.0 - 0.05));
      sigma = std::min(std::max(sigma, 0.05), 0.25);
    }
    out[i] = sigma;
  }
}
API double orc_dds_quadratic(int ndim, const double *xmin, const double *xmax, const double *xstart, const double *center, int nfunc_max, int num_threads, double sigma, double *xbest, double *trace_cost) {
  DDS d; d.ndim = ndim; d.pb.resize(ndim); d.nfunc_max = nfunc_max; d.num_threads = num_threads; d.sigma_init = sigma;
  std::vector<double> xs(ndim);
  for (int i = 0; i < ndim; i++) { d.pb[i] = {xmin[i], xmax[i]}; xs[i] = xstart[i]; }
  int ne = 0;
  auto f = [&](const std::vector<double> &x) {
    double s = 0;
    for (int i = 0; i < ndim; i++) { double dd = x[i] - center[i]; s += (i + 1) * dd * dd; }
    if (trace_cost && ne < nfunc_max) trace_cost[ne] = s;
    ne++;
    return s;
  };
  auto r = d.run(f, xs);
  for (int i = 0; i < ndim; i++) xbest[i] = r.second[i];
  return r.first;
}
API int orc_encode_frame(int nch, int framesize, int n, const int32_t *raw, const orc_frame_cfg *rc, float *profile_io, uint8_t *out, int cap, double *trace_cost, float *trace_coefs, int *info) {
  std::vector<uint8_t> rec;
  int len = encode_frame(nch, framesize, n, raw, *rc, profile_io, rec, trace_cost, trace_coefs, info);
  if (len > cap) return -len;
  std::memcpy(out, rec.data(), len);
  return len;
}
API int orc_decode_frame(const uint8_t *rec, int len, int nch, int framesize, int32_t *out, int cap, float *coefs_out) {
  return decode_frame(rec, len, nch, framesize, out, cap, coefs_out);
}
API double orc_dot(const double *x, const double *y, int n) { return dot_ref(x, y, n); }
API double orc_s2pow(const double *x, const double *p, int n) { return s2pow_ref(x, p, n); }
API int orc_ldlt(const double *A, int n, double nu, const double *b, double *w) {
  LDLT l; l.init(n);
  int ok = l.factor(A, nu);
  if (ok) l.solve(b, w);
  return ok;
}
API void orc_laplace_table(int bpn, int count, uint16_t *out) {
  for (int a = 0; a < count; a++) out[a] = (uint16_t)Bitplane::laplace((uint32_t)a, bpn);
}
API int orc_abi_version(void) { return 1; }
