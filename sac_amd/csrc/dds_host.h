// sac_amd/csrc/dds_host.h -- host-side DDS search state for one frame (no HIP dependency).
//
// Reference: OptDDS (/root/reference/src/opt/dds.cpp:12-119), Opt::gen_norm / reflect
// (opt/opt.cpp:111-116,156-166), SSC0 / SSC1 (opt/ssc.h), Random (common/rand.h).
// Compile with -ffp-contract=off: the fused multiply-adds of the reference build are explicit.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

namespace sacamd {
struct BoxCoef { float vmin, vmax, vdef; };
}
using Coef = sacamd::BoxCoef;

namespace {

// Opt::rand (common/rand.h) seeded 0 per optimiser instance (opt.cpp:5), one optimiser per frame (libsac.cpp:410),
// and Opt::reflect; shared by the three searchers.
struct SearchRng {
  std::mt19937 eng{0};
  double r01() { return std::uniform_real_distribution<double>{0, 1}(eng); }
  // std::normal_distribution<double>{0,1} constructed per draw (rand.h:25-27) == one polar-method
  // round returning y*mult; x*x+y*y is a fused multiply-add in the reference build.
  double rnorm() {
    double x, y, r2;
    do { x = 2.0 * r01() - 1.0; y = 2.0 * r01() - 1.0; r2 = std::fma(x, x, y * y); } while (r2 > 1.0 || r2 == 0.0);
    const double mult = std::sqrt(-2 * std::log(r2) / r2);
    return y * mult;
  }
  unsigned ruint(unsigned a, unsigned b) { return std::uniform_int_distribution<uint32_t>{a, b}(eng); }
  static double reflect(double x, double lo, double hi) {   // opt.cpp:156-166
    if (x < lo) { x = lo + (lo - x); if (x > hi) x = lo; }
    if (x > hi) { x = hi - (x - hi); if (x < lo) x = hi; }
    return x;
  }
};

struct FrameSearch : SearchRng {
  std::vector<double> xb; double cb = 0.0;   // best point / cost
  double sigma = 0.2;
  int nfunc = 1;
  int nsucc = 0, nfail = 0;                  // SSC0
  double p_succ = 0.05;                      // SSC1(0.05,0.10,0.05)
  std::vector<std::vector<double>> gen;      // candidates of the current generation
  std::vector<double> gcost;

  std::vector<double> candidate(const Coef *box, const std::vector<int> &lp, int nfunc_max) {   // dds.cpp:12-30
    const int ndim = (int)lp.size();
    std::vector<int> J;
    const double p = 1.0 - std::log((double)nfunc) / std::log((double)nfunc_max);
    for (int i = 0; i < ndim; i++) if (r01() < p) J.push_back(i);
    if (J.empty()) J.push_back((int)ruint(0, ndim - 1));
    std::vector<double> xt = xb;
    for (int k : J) {
      const double lo = box[lp[k]].vmin, hi = box[lp[k]].vmax;
      const double sg = sigma * (hi - lo);
      xt[k] = reflect(std::fma(sg, rnorm(), xb[k]), lo, hi);   // gen_norm, opt.cpp:111-116
    }
    return xt;
  }
  // run_single selection + SSC0(3,50) (dds.cpp:44-55, ssc.h:14-32)
  void select_single(double cost) {
    double lam = 0.0;
    if (cost < cb) { cb = cost; xb = gen[0]; lam = 1.0; }
    if (lam > 0.0) { nsucc++; nfail = 0; } else { nsucc = 0; nfail++; }
    if (nsucc >= 3) { sigma *= 2.0; nsucc = 0; } else if (nfail >= 50) { sigma /= 2.0; nfail = 0; }
    sigma = std::min(std::max(sigma, 0.05), 0.5);
  }
  // run_mt selection + SSC1(0.05,0.10,0.05) (dds.cpp:86-97, ssc.h:43-55)
  void select_mt(const double *gc, int nt) {
    const double old = cb; int ns = 0;
    for (int i = 0; i < nt; i++) if (gc[i] < old) { ns++; if (gc[i] < cb) { cb = gc[i]; xb = gen[i]; } }
    const double lam = ns / static_cast<double>(nt);
    p_succ = std::fma(1.0 - 0.10, p_succ, 0.10 * lam);
    sigma = sigma * std::exp(0.05 * (p_succ - 0.05) / (1.0 - 0.05));
    sigma = std::min(std::max(sigma, 0.05), 0.25);
  }
};

}  // namespace

