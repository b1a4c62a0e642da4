// sac_amd/csrc/search_host.h -- host-side state of the two other searchers of FrameCoder::Optimize (libsac.cpp:408-415),
// one instance per frame, driven in lock-step generations over the batched objective (sacamd_evaluate):
//   FrameSearchDE   == OptDE  (/root/reference/src/opt/de.cpp:10-184, opt/de.h; JADE-style current-to-pbest/1/bin)
//   FrameSearchCMA  == OptCMA (opt/cma.cpp:6-92, opt/cma.h; (1+1)-CMA-ES with slmath::Cholesky, common/math.h:82-127)
// A searcher hands out the candidates of its next generation (propose) and takes their costs (accept); which frames
// and candidates share a kernel launch is the caller's business.  No HIP dependency.
// Compile with -ffp-contract=off: the fused multiply-adds of the reference build are explicit.
#pragma once
#include <algorithm>
#include <cmath>
#include <numeric>
#include <utility>
#include <vector>

#include "canon.h"
#include "dds_host.h"

namespace {

struct SearchBox { std::vector<double> lo, hi; };

// ---------------------------------------------------------------------------------------------------- DE
struct FrameSearchDE : SearchRng {
  // OptDE::DECfg defaults (de.h:19-31); sigma_init / nfunc_max come from the command line (cmdline.cpp:227-231)
  static constexpr int NP = 30, NPBEST = 2;            // npbest = clamp(round(0.1 * 30) - 1, 0, 29)
  static constexpr double CR0 = 0.5, F0 = 0.5, C = 0.1;
  struct Agent { double cost; std::vector<double> x; };
  const SearchBox *box = nullptr;
  int ndim = 0, nfunc_max = 0;
  double sigma_init = 0.15;
  std::vector<Agent> pop;
  Agent best;
  double mCR = CR0, mF = F0;
  int nfunc = 1;
  bool initialised = false;
  std::vector<std::vector<double>> gen;                // trial vectors of the current generation
  std::vector<std::pair<double, double>> gen_mut;      // their (CR, F)

  void start(const SearchBox *b, const std::vector<double> &x0, double c0, int nmax, double sigma) {
    box = b; ndim = (int)x0.size(); nfunc_max = nmax; sigma_init = sigma;
    best = Agent{c0, x0};
    pop.assign(NP, Agent{0.0, {}});
    pop[0] = best;
  }
  // normal_distribution{mu, sd}: ret * sd + mu, a fused multiply-add in the reference build
  double rnorm2(double mu, double sd) { return std::fma(rnorm(), sd, mu); }
  // libstdc++ cauchy_distribution (bits/random.tcc): a + b * tan(pi * u), u != 0.5
  double rcauchy(double a, double b) {
    double u;
    do u = r01(); while (u == 0.5);
    const double pi = 3.1415926535897932384626433832795029L;
    return a + b * std::tan(pi * u);
  }
  std::vector<int> pick_except(int n, int except, int k) {     // select_k_unique_except, de.cpp:12-29
    std::vector<int> r;
    if (k >= n - 1) return r;
    std::vector<int> e;
    for (int i = 0; i < n; i++) if (i != except) e.push_back(i);
    for (int i = 0; i < k; i++) {
      const int idx = (int)ruint(0, (unsigned)e.size() - 1);
      r.push_back(e[idx]);
      e.erase(e.begin() + idx);
    }
    return r;
  }
  std::vector<double> trial(int ia) {                           // generate_candidate, de.cpp:31-70 (CURPBEST)
    const double tCR = std::min(std::max(rnorm2(mCR, 0.1), 0.01), 1.0);
    const double tF = std::min(std::max(rcauchy(mF, 0.1), 0.01), 1.0);
    const int R = (int)ruint(0, ndim - 1);
    const std::vector<int> v = pick_except((int)pop.size(), ia, 2);
    const int np = std::min(NPBEST, (int)pop.size() - 1);
    const int xp = np > 0 ? (int)ruint(0, np) : 0;
    const std::vector<double> &pb = pop[xp].x, &cur = pop[ia].x, &x1 = pop[v[0]].x, &x2 = pop[v[1]].x;
    std::vector<double> xt(ndim);
    for (int i = 0; i < ndim; i++) {
      // mut_curbest, de.cpp:176-184: cur + F*(pbest - cur) + F*(x1 - x2), two fused multiply-adds
      const double y = std::fma(tF, x1[i] - x2[i], std::fma(tF, pb[i] - cur[i], cur[i]));
      const double xm = reflect(y, box->lo[i], box->hi[i]);
      xt[i] = (r01() < tCR || i == R) ? xm : cur[i];
    }
    gen_mut.push_back({tCR, tF});
    return xt;
  }
  // candidates of the next generation; empty = search finished
  const std::vector<std::vector<double>> &propose() {
    gen.clear(); gen_mut.clear();
    if (!initialised) {                                         // random population around the start point, de.cpp:88-100
      for (int a = 1; a < NP; a++) {
        std::vector<double> xt(ndim);
        for (int i = 0; i < ndim; i++) {                        // gen_norm_samples, opt.cpp:111-116,125-133
          const double sg = sigma_init * (box->hi[i] - box->lo[i]);
          xt[i] = reflect(std::fma(sg, rnorm(), best.x[i]), box->lo[i], box->hi[i]);
        }
        gen.push_back(std::move(xt));
      }
      return gen;
    }
    if (nfunc >= nfunc_max) return gen;
    std::sort(pop.begin(), pop.end(), [](const Agent &a, const Agent &b) { return a.cost < b.cost; });   // de.cpp:118-121
    const int agents = std::min(nfunc_max - nfunc, (int)pop.size());
    for (int ia = 0; ia < agents; ia++) gen.push_back(trial(ia));
    return gen;
  }
  void accept(const double *cost) {
    const int n = (int)gen.size();
    nfunc += n;
    if (!initialised) {
      for (int a = 1; a < NP; a++) { pop[a] = Agent{cost[a - 1], gen[a - 1]}; }
      for (int a = 1; a < NP; a++) if (pop[a].cost < best.cost) best = pop[a];
      initialised = true;
      return;
    }
    std::vector<double> crs, fs;                                // greedy selection, de.cpp:141-152
    for (int ia = 0; ia < n; ia++)
      if (cost[ia] < pop[ia].cost) {
        pop[ia] = Agent{cost[ia], gen[ia]};
        crs.push_back(gen_mut[ia].first); fs.push_back(gen_mut[ia].second);
        if (pop[ia].cost < best.cost) best = pop[ia];
      }
    if (nfunc >= nfunc_max) return;
    double mean = 0.0;                                          // MathUtils::mean / meanL, utils.h:283-305
    if (!crs.empty()) { double s = 0.0; for (double v : crs) s += v; mean = s / static_cast<double>(crs.size()); }
    double lehmer = 0.0;
    if (!fs.empty()) {
      // as the reference build runs the loop: blocks of four with the squares formed first (vectorised) and added
      // in order, the last one to three elements fused
      double s0 = 0.0, s1 = 0.0;
      size_t k = 0;
      for (; k + 4 <= fs.size(); k += 4)
        for (size_t q = k; q < k + 4; q++) { s0 = s0 + fs[q] * fs[q]; s1 += fs[q]; }
      for (; k < fs.size(); k++) { s0 = std::fma(fs[k], fs[k], s0); s1 += fs[k]; }
      if (s1 > 0.0) lehmer = s0 / s1;
    }
    mCR = std::fma(mean, C, (1.0 - C) * mCR);                   // de.cpp:158-159 as contracted by the reference build
    mF = std::fma(lehmer, C, (1.0 - C) * mF);
  }
  bool done() const { return initialised && nfunc >= nfunc_max; }
};

// ---------------------------------------------------------------------------------------------------- CMA
struct FrameSearchCMA : SearchRng {
  const SearchBox *box = nullptr;
  int ndim = 0, nfunc_max = 0, nfunc = 1;
  // CMAParams, cma.h:17-38
  double d = 0, p_target = 2.0 / 11.0, cp = 1.0 / 12.0, cc = 0, ccov = 0, sigma = 0, p_succ = 0;
  std::vector<double> pc, mcov, G, az, xbest;                   // mcov, G: ndim x ndim row-major
  double cbest = 0.0;
  std::vector<std::vector<double>> gen;

  void start(const SearchBox *b, const std::vector<double> &x0, double c0, int nmax, double sigma_init) {
    box = b; ndim = (int)x0.size(); nfunc_max = nmax;
    const int n = ndim;
    d = 1.0 + n / 2.0; cc = 2.0 / (n + 2.0); ccov = 2.0 / (n * n + 6.0);
    sigma = sigma_init; p_succ = p_target;                      // SSC1(p_target, cp, 1/d): p_succ starts at the target
    pc.assign(n, 0.0); mcov.assign((size_t)n * n, 0.0); G.assign((size_t)n * n, 0.0); az.assign(n, 0.0);
    for (int i = 0; i < n; i++) mcov[(size_t)i * n + i] = 1.0;
    xbest = x0; cbest = c0;
  }
  // in-order reduction sum -= a[k]*b[k] as the vectorised loop of the reference build runs it (canon.h fold_add, negated)
  static double fold_sub(double acc, int m, const double *a, const double *b) {
    int k = 0;
    for (; k + 4 <= m; k += 4) { acc = acc - a[k] * b[k]; acc = acc - a[k + 1] * b[k + 1]; acc = acc - a[k + 2] * b[k + 2]; acc = acc - a[k + 3] * b[k + 3]; }
    if (m - k >= 2) { acc = acc - a[k] * b[k]; acc = acc - a[k + 1] * b[k + 1]; k += 2; }
    if (k < m) acc = std::fma(-a[k], b[k], acc);
    return acc;
  }
  void factor() {                                               // slmath::Cholesky::Factor(mcov, 0.1), math.h:89-111
    const int n = ndim;
    for (int i = 0; i < n; i++) std::copy_n(&mcov[(size_t)i * n], i + 1, &G[(size_t)i * n]);
    for (int i = 0; i < n; i++) {
      double *gi = &G[(size_t)i * n];
      for (int j = 0; j < i; j++) { const double *gj = &G[(size_t)j * n]; gi[j] = fold_sub(gi[j], j, gi, gj) / gj[j]; }
      const double s = fold_sub(gi[i] + 0.1, i, gi, gi);
      if (s > 1E-8) gi[i] = std::sqrt(s); else return;          // the reference ignores the failure and uses G as it stands
    }
  }
  const std::vector<std::vector<double>> &propose() {
    gen.clear();
    if (nfunc >= nfunc_max) return gen;
    factor();
    const int n = ndim;
    std::vector<double> z(n), x(n);
    for (double &r : z) r = rnorm();
    for (int i = 0; i < n; i++) az[i] = sacamd::dot_canon(&G[(size_t)i * n], z.data(), n);    // slmath::mul(G, z)
    for (int i = 0; i < n; i++) {
      const double scale = (box->hi[i] - box->lo[i]) * sigma;
      x[i] = reflect(std::fma(scale, az[i], xbest[i]), box->lo[i], box->hi[i]);
    }
    gen.push_back(std::move(x));
    return gen;
  }
  void accept(const double *cost) {
    const double fn = cost[0];
    const double lambda = fn < cbest ? 1.0 : 0.0;
    p_succ = std::fma(1.0 - cp, p_succ, cp * lambda);           // SSC1::update, ssc.h:50-56 (bounds 0.05 .. 0.25)
    sigma = sigma * std::exp((1.0 / d) * (p_succ - p_target) / (1.0 - p_target));
    sigma = std::min(std::max(sigma, 0.05), 0.25);
    if (fn < cbest) {
      cbest = fn; xbest = gen[0];
      const int n = ndim;
      const double a = 1.0 - cc, b = std::sqrt(cc * (2.0 - cc));
      for (int i = 0; i < n; i++) pc[i] = std::fma(a, pc[i], b * az[i]);                        // update_cov, cma.cpp:45-49
      const double c1 = 1.0 - ccov;
      for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) mcov[(size_t)j * n + i] = std::fma(c1, mcov[(size_t)j * n + i], ccov * (pc[j] * pc[i]));
    }
    nfunc++;
  }
  bool done() const { return nfunc >= nfunc_max; }
};

}  // namespace
