// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked or loaded by the product).
//
// A thin C-ABI over the *genuine* reference classes, compiled by oracle/Makefile from the
// sources where they lie under /root/reference/src (no copies, no stand-in headers) into
// oracle/_ref/libsacref.so.  Only the reference translation units that build with the
// image's g++ 11.4 are used (the ones that do not need <format>):
//   libsac/{libsac,pred,profile,vle,map}.cpp pred/{ols,rls}.cpp model/range.cpp
//   common/{utils,md5}.cpp opt/opt.cpp file/{file,sac}.cpp
// opt/dds.cpp, file/wav.cpp and cmdline.cpp need <format> and are therefore NOT built;
// FrameCoder::Predict/Optimize (which reference OptDDS) are dropped by --gc-sections and
// their ~60 lines of control flow are restated below (marked RESTATED) around the genuine
// FrameCoder::PredictFrame / GetCost / CnvError_S2U / Encode / WriteEncoded / ReadEncoded /
// Decode / Unpredict, Predictor, BitplaneCoder, RangeCoderSH, Remap, MapEncoder, Cost*,
// SacProfile, Opt (RNG, gen_norm, reflect) and SSC0/SSC1.
//
// This TU is compiled with -fno-access-control so it can call private members.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>
#include <string>
#include <cmath>

#include "libsac/libsac.h"
#include "libsac/pred.h"
#include "libsac/cost.h"
#include "libsac/vle.h"
#include "libsac/map.h"
#include <numeric>
#include <limits>
#include <stdexcept>
#include "common/utils.h"
#include "libsac/sparse.h"
#include "file/sac.h"
#include "opt/opt.h"
#include "opt/ssc.h"
#include "opt/de.h"
#include "opt/cma.h"
#include "common/math.h"
#include <memory>
#include <atomic>
#include <span>
#include <tuple>

#define API extern "C" __attribute__((visibility("default")))

namespace {

struct OptProbe : public Opt {
  using Opt::Opt;
  ppoint run(opt_func, const vec1D &x) override { return {0.0, x}; }
};

// RESTATED from opt/dds.cpp:12-119 (that file needs <format>); uses the genuine Opt base
// (Random, gen_norm, reflect) and the genuine SSC0/SSC1.
bool g_parallel_eval = false;   // ref_set_parallel_eval: candidates of a DDS generation on their own threads, as the reference runs them

struct DriverDDS : public Opt {
  OptDDS::DDSCfg cfg;
  std::vector<double> *trace_cost = nullptr;
  std::vector<vec1D> *trace_x = nullptr;
  DriverDDS(const OptDDS::DDSCfg &c, const box_const &pb) : Opt(pb), cfg(c) {}

  vec1D generate_candidate(const vec1D &x, int nfunc, double sigma) {
    std::vector<int> J;
    double p = 1.0 - log(nfunc) / log(cfg.nfunc_max);
    for (int i = 0; i < ndim; i++)
      if (rand.event(p)) J.push_back(i);
    if (!J.size()) J.push_back(rand.ru_int(0, ndim - 1));
    vec1D xtest = x;
    for (auto k : J) xtest[k] = gen_norm(x[k], pb[k], sigma);
    return xtest;
  }
  double eval(opt_func &func, const vec1D &x) {
    double c = func(x);
    if (trace_cost) trace_cost->push_back(c);
    if (trace_x) trace_x->push_back(x);
    return c;
  }
  ppoint run_single(opt_func func, const vec1D &xstart) {
    int nfunc = 1;
    ppoint xb{eval(func, xstart), xstart};
    double sigma = cfg.sigma_init;
    SSC0 ssc(cfg.c_succ_max, cfg.c_fail_max);
    while (nfunc < cfg.nfunc_max) {
      ppoint x_gen;
      x_gen.second = generate_candidate(xb.second, nfunc, sigma);
      x_gen.first = eval(func, x_gen.second);
      nfunc++;
      double lambda = 0.0;
      if (x_gen.first < xb.first) { xb = x_gen; lambda = 1.0; }
      sigma = ssc.update(sigma, lambda);
    }
    return xb;
  }
  ppoint run_mt(opt_func func, const vec1D &xstart) {
    ppoint xb{eval(func, xstart), xstart};
    double sigma = cfg.sigma_init;
    SSC1 ssc(0.05, 0.10, 0.05);
    int nfunc = 1;
    while (nfunc < cfg.nfunc_max) {
      const int nthreads = std::min(cfg.nfunc_max - nfunc, cfg.num_threads);
      opt_points x_gen(nthreads);
      for (int i = 0; i < nthreads; i++) {
        x_gen[i].second = generate_candidate(xb.second, nfunc, sigma);
        nfunc++;
      }
      // evaluation order does not matter for the result (pure function): serial by default (traces), or -- for timing the
      // reference's own threading -- the genuine Opt::eval_points_mt (one std::async per candidate, opt/opt.cpp:11-43)
      if (g_parallel_eval) eval_points_mt(func, std::span<ppoint>(x_gen));
      else for (int i = 0; i < nthreads; i++) x_gen[i].first = eval(func, x_gen[i].second);
      ppoint xb_old = xb;
      int nsucc = 0;
      for (const auto &xg : x_gen)
        if (xg.first < xb_old.first) {
          nsucc++;
          if (xg.first < xb.first) xb = xg;
        }
      double lambda = nsucc / static_cast<double>(nthreads);
      sigma = ssc.update(sigma, lambda);
    }
    return xb;
  }
  ppoint run(opt_func func, const vec1D &xstart) override {
    if (cfg.num_threads <= 0) return run_single(func, xstart);
    return run_mt(func, xstart);
  }
};

// RESTATED from opt/de.cpp:10-184 (needs <format>): OptDE's population loop around the genuine Opt base
// (Random, gen_norm_samples, reflect) and the genuine DECfg (opt/de.h); evaluation is serial (eval_pop_pool's
// result does not depend on the thread count: func is pure).
struct DriverDE : public Opt {
  OptDE::DECfg cfg;
  DriverDE(const OptDE::DECfg &c, const box_const &pb) : Opt(pb), cfg(c) {}
  double gen_CR(double mCR) { return std::clamp(rand.r_norm(mCR, 0.1), 0.01, 1.0); }
  double gen_F(double mF) { return std::clamp(rand.r_cauchy(mF, 0.1), 0.01, 1.0); }
  std::vector<int> select_k_unique_except(int n, int ie, int k) {
    std::vector<int> r;
    if (k >= n - 1) return r;
    std::vector<int> e(n);
    std::iota(std::begin(e), std::end(e), 0);
    std::erase(e, ie);
    for (int i = 0; i < k; i++) {
      int idx = rand.ru_int(0, e.size() - 1);
      int val = e[idx];
      r.push_back(val);
      std::erase(e, val);
    }
    return r;
  }
  vec1D mut_curbest(const vec1D &xbest, const vec1D &xb, const vec1D &x1, const vec1D &x2, double F) {
    vec1D xm(ndim);
    for (int i = 0; i < ndim; i++) {
      double y = xb[i] + F * (xbest[i] - xb[i]) + F * (x1[i] - x2[i]);
      xm[i] = reflect(y, pb[i].xmin, pb[i].xmax);
    }
    return xm;
  }
  vec1D mut_1bin(const vec1D &xb, const vec1D &x1, const vec1D &x2, double F) {
    vec1D xm(ndim);
    for (int i = 0; i < ndim; i++) {
      double y = xb[i] + F * (x1[i] - x2[i]);
      xm[i] = reflect(y, pb[i].xmin, pb[i].xmax);
    }
    return xm;
  }
  auto generate_candidate(const opt_points &pop, const vec1D &xbest, int iagent, double mCR, double mF) {
    const double tCR = gen_CR(mCR);
    const double tF = gen_F(mF);
    const int R = rand.ru_int(0, ndim - 1);
    auto gp = [&](int i) -> auto & { return pop[i].second; };
    const int mutvals = cfg.mut_method == OptDE::RAND1BIN ? 3 : 2;     // OptDE::MutVals (opt/de.h:10-15)
    auto v = select_k_unique_except(pop.size(), iagent, mutvals);
    vec1D xm;
    if (cfg.mut_method == OptDE::BEST1BIN) xm = mut_1bin(xbest, gp(v[0]), gp(v[1]), tF);
    else if (cfg.mut_method == OptDE::RAND1BIN) xm = mut_1bin(gp(v[0]), gp(v[1]), gp(v[2]), tF);
    else if (cfg.mut_method == OptDE::CUR1BEST) xm = mut_curbest(xbest, gp(iagent), gp(v[0]), gp(v[1]), tF);
    else {
      int np = std::min(cfg.npbest, static_cast<int>(pop.size()) - 1);
      int xp = np > 0 ? rand.ru_int(0, np) : 0;
      xm = mut_curbest(gp(xp), gp(iagent), gp(v[0]), gp(v[1]), tF);
    }
    vec1D xtrial(ndim);
    const ppoint &xi = pop[iagent];
    for (int i = 0; i < ndim; i++) {
      if (rand.event(tCR) || (i == R)) xtrial[i] = xm[i];
      else xtrial[i] = xi.second[i];
    }
    return std::tuple{xtrial, tCR, tF};
  }
  ppoint run(opt_func func, const vec1D &xstart) override {
    std::size_t nfunc = 1;
    ppoint xb{func(xstart), xstart};
    opt_points pop(cfg.NP);
    pop[0] = xb;
    std::span<ppoint> pop_span(pop.begin() + 1, pop.end());
    for (auto &x : pop_span) {
      vec1D xt;
      if (cfg.init_method == OptDE::INIT_UNIV) xt = gen_uniform_samples(xb.second, cfg.sigma_init);
      else xt = gen_norm_samples(xb.second, cfg.sigma_init);
      x.second = xt;
    }
    for (auto &x : pop_span) { x.first = func(x.second); nfunc++; }
    for (const auto &x : pop_span)
      if (x.first < xb.first) xb = x;
    double mCR = cfg.CR, mF = cfg.F;
    opt_points gen_pop;
    std::vector<std::pair<double, double>> gen_mut;
    while (nfunc < cfg.nfunc_max) {
      if (cfg.mut_method == OptDE::CURPBEST)
        std::sort(begin(pop), end(pop), [](const auto &a, const auto &b) { return a.first < b.first; });
      const int num_agents = std::min(cfg.nfunc_max - nfunc, pop.size());
      gen_mut.resize(num_agents);
      gen_pop.resize(num_agents);
      for (int iagent = 0; iagent < num_agents; iagent++) {
        auto [xtrial, tCR, tF] = generate_candidate(pop, xb.second, iagent, mCR, mF);
        gen_mut[iagent] = {tCR, tF};
        gen_pop[iagent].second = xtrial;
      }
      for (auto &x : gen_pop) { x.first = func(x.second); nfunc++; }
      std::vector<double> CR_succ, F_succ;
      for (int iagent = 0; iagent < num_agents; iagent++)
        if (gen_pop[iagent].first < pop[iagent].first) {
          pop[iagent] = gen_pop[iagent];
          CR_succ.push_back(gen_mut[iagent].first);
          F_succ.push_back(gen_mut[iagent].second);
          if (pop[iagent].first < xb.first) xb = pop[iagent];
        }
      if (nfunc >= cfg.nfunc_max) break;
      mCR = (1.0 - cfg.c) * mCR + cfg.c * MathUtils::mean(CR_succ);
      mF = (1.0 - cfg.c) * mF + cfg.c * MathUtils::meanL(F_succ);
    }
    return xb;
  }
};

// RESTATED from opt/cma.cpp:6-92 (needs <format>): the (1+1)-CMA loop around the genuine slmath::Cholesky / mul /
// mul_add / outer (common/math.h), SSC1 (opt/ssc.h), CMAParams (opt/cma.h) and Opt base.
struct DriverCMA : public Opt {
  OptCMA::CMACfg cfg;
  OptCMA::CMAParams p;
  slmath::Cholesky chol;
  DriverCMA(const OptCMA::CMACfg &c, const box_const &pb) : Opt(pb), cfg(c), p(ndim), chol(ndim) {
    p.sigma = cfg.sigma_init;
    p.psucc = p.p_target_succ;
  }
  auto generate_candidate(const vec1D &x, double sigma) {
    vec1D z(ndim);
    for (auto &r : z) r = rand.r_norm();
    vec1D az = slmath::mul(chol.G, z);
    vec1D xgen(ndim);
    for (int i = 0; i < ndim; i++) {
      double scale = (pb[i].xmax - pb[i].xmin) * sigma;
      double xnew = x[i] + scale * az[i];
      xgen[i] = reflect(xnew, pb[i].xmin, pb[i].xmax);
    }
    return std::tuple{xgen, az};
  }
  void update_cov(vec2D &mcov, vec1D &pc, const vec1D &az) {
    pc = slmath::mul_add(1.0 - p.cc, pc, std::sqrt(p.cc * (2.0 - p.cc)), az);
    mcov = slmath::mul_add(1.0 - p.ccov, mcov, p.ccov, slmath::outer(pc, pc));
  }
  ppoint run(opt_func func, const vec1D &xstart) override {
    vec1D pc(ndim);
    vec2D mcov(ndim, vec1D(ndim));
    for (int i = 0; i < ndim; i++) mcov[i][i] = 1.0;
    SSC1 ssc(p.p_target_succ, p.cp, 1.0 / p.d);
    int nfunc = 1;
    ppoint xb{func(xstart), xstart};
    while (nfunc < cfg.nfunc_max) {
      chol.Factor(mcov, 0.1);
      auto [xgen, az] = generate_candidate(xb.second, p.sigma);
      double fn = func(xgen);
      double lambda = (fn < xb.first) ? 1.0 : 0.0;
      p.sigma = ssc.update(p.sigma, lambda);
      if (fn < xb.first) {
        xb.first = fn;
        xb.second = xgen;
        update_cov(mcov, pc, az);
      }
      nfunc++;
    }
    return xb;
  }
};

// the searcher FrameCoder::Optimize would construct (libsac.cpp:408-415) with the settings cmdline.cpp:221-241 derives
std::unique_ptr<Opt> make_searcher(int search, int maxnfunc, int num_threads, double sigma, const Opt::box_const &pb) {
  if (search == 1) {
    OptDE::DECfg c;
    c.nfunc_max = maxnfunc; c.num_threads = std::max(num_threads, 1); c.sigma_init = sigma;
    return std::make_unique<DriverDE>(c, pb);
  }
  if (search == 2) {
    OptCMA::CMACfg c;
    c.nfunc_max = maxnfunc; c.num_threads = std::max(num_threads, 1); c.sigma_init = sigma;
    return std::make_unique<DriverCMA>(c, pb);
  }
  OptDDS::DDSCfg c;
  c.nfunc_max = maxnfunc; c.num_threads = num_threads; c.sigma_init = sigma;
  return std::make_unique<DriverDDS>(c, pb);
}
int g_search_method = 0;   // FrameCoder::SearchMethod for the next ref_encode_frame calls: 0 DDS, 1 DE, 2 CMA

FrameCoder::tsac_cfg make_cfg(int optk, int sparse_pcm, int zero_mean) {
  FrameCoder::tsac_cfg cfg;
  cfg.ocfg.optk = optk;
  cfg.sparse_pcm = sparse_pcm;
  cfg.zero_mean = zero_mean;
  return cfg;
}

void set_profile(SacProfile &p, const float *coefs) {
  for (size_t i = 0; i < p.coefs.size(); i++) p.coefs[i].vdef = coefs[i];
}

std::string tmpname() {
  char buf[] = "/tmp/sacref_XXXXXX";
  int fd = mkstemp(buf);
  if (fd >= 0) close(fd);
  return std::string(buf);
}

} // namespace

// ---------------------------------------------------------------- profile / tables
API int ref_profile(float *out) {
  SacProfile p;
  int n = p.LoadBaseProfile();
  for (int i = 0; i < n; i++) {
    out[3 * i + 0] = p.coefs[i].vmin;
    out[3 * i + 1] = p.coefs[i].vmax;
    out[3 * i + 2] = p.coefs[i].vdef;
  }
  return n;
}

API void ref_domain_tables(int *fwd /*32768*/, int *inv /*4095: x=-2047..2047*/) {
  for (int i = 0; i < PSCALE; i++) fwd[i] = myDomain.Fwd(i);
  for (int x = -2047; x <= 2047; x++) inv[x + 2047] = myDomain.Inv(x);
}

// ---------------------------------------------------------------- predictor
// genuine FrameCoder::PredictFrame (libsac.cpp:94-142). samples are mean-removed, planar
// [nch][total]; stats = per channel {minval,maxval,mean} (already shifted by mean).
API int ref_predict_frame(int nch, int framesize, int total, const int32_t *samples,
                          const int32_t *stats, const float *coefs, int from, int n,
                          int optimize, int optk, int32_t *error, int32_t *pred) {
  FrameCoder fc(nch, framesize, make_cfg(optk, 1, 1));
  fc.SetNumSamples(total);
  for (int ch = 0; ch < nch; ch++) {
    std::copy_n(samples + (size_t)ch * total, total, fc.samples[ch].begin());
    fc.framestats[ch].minval = stats[3 * ch + 0];
    fc.framestats[ch].maxval = stats[3 * ch + 1];
    fc.framestats[ch].mean = stats[3 * ch + 2];
  }
  SacProfile prof = fc.base_profile;
  set_profile(prof, coefs);
  FrameCoder::tch_samples err(nch, std::vector<int32_t>(n));
  fc.PredictFrame(prof, err, from, n, optimize != 0);
  for (int ch = 0; ch < nch; ch++) {
    std::copy_n(err[ch].begin(), n, error + (size_t)ch * n);
    if (pred && !optimize) std::copy_n(fc.pred[ch].begin(), n, pred + (size_t)ch * n);
  }
  return 0;
}

// Per-step trace with the genuine Predictor; the sample loop of libsac.cpp:104-141 is
// RESTATED here only to expose pd / p_lpc / p_lms. Outputs are indexed by *file* channel.
API int ref_predict_trace(int nch, int total, const int32_t *samples, const int32_t *stats,
                          const float *coefs, int from, int n, int optimize, int optk,
                          double *pd_out, double *plpc_out, double *plms_out,
                          int32_t *error) {
  FrameCoder fc(nch, 16, make_cfg(optk, 1, 1));
  SacProfile prof = fc.base_profile;
  set_profile(prof, coefs);
  Predictor::tparam param;
  fc.SetParam(param, prof, optimize != 0);
  int32_t mn[2], mx[2];
  for (int ch = 0; ch < nch; ch++) { mn[ch] = stats[3 * ch]; mx[ch] = stats[3 * ch + 1]; }
  Range r0{.lo = mn[0], .hi = mx[0]};
  Range r1 = r0;
  if (nch == 2) r1 = {.lo = mn[1], .hi = mx[1]};
  Predictor pr(r0, r1, param);
  auto eprocess = [&](int ch_p, int ch, int32_t val, int idx) {
    double pd = pr.predict(ch_p);
    int32_t pi = std::clamp((int32_t)std::round(pd), mn[ch], mx[ch]);
    pd_out[(size_t)ch * n + idx] = pd;
    plpc_out[(size_t)ch * n + idx] = pr.p_lpc[ch_p];
    plms_out[(size_t)ch * n + idx] = pr.p_lms[ch_p];
    error[(size_t)ch * n + idx] = val - pi;
    pr.update(ch_p, val);
  };
  if (nch == 1) {
    const int32_t *src = samples + from;
    for (int idx = 0; idx < n; idx++) {
      pr.fillbuf_ch0(src, idx, src, idx);
      eprocess(0, 0, src[idx], idx);
    }
  } else {
    int ch0 = param.ch_ref, ch1 = 1 - ch0;
    const int32_t *src0 = samples + (size_t)ch0 * total + from;
    const int32_t *src1 = samples + (size_t)ch1 * total + from;
    int idx0 = 0, idx1 = 0;
    while (idx0 < n || idx1 < n) {
      if (idx0 < n) {
        pr.fillbuf_ch0(src0, idx0, src1, idx1);
        eprocess(0, ch0, src0[idx0], idx0);
        idx0++;
      }
      if (idx0 >= param.nS1) {
        pr.fillbuf_ch1(src0, src1, idx1, n);
        eprocess(1, ch1, src1[idx1], idx1);
        idx1++;
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------- cost functions
API double ref_cost(int kind, const int32_t *buf, int n) {
  std::span<const int32_t> s{buf, (size_t)n};
  switch (kind) {
    case 0: return CostL1().Calc(s);
    case 1: return CostRMS().Calc(s);
    case 2: return CostEntropy().Calc(s);
    case 3: return CostGolomb().Calc(s);
    case 4: return CostBitplane().Calc(s);
  }
  return -1.0;
}

// ---------------------------------------------------------------- bitplane / range coder
API int ref_bitplane_encode(const int32_t *s2u, int n, int maxbpn, uint8_t *out, int cap) {
  BufIO buf;
  RangeCoderSH rc(buf);
  rc.Init();
  BitplaneCoder bc(maxbpn, n);
  std::vector<int32_t> tmp(s2u, s2u + n);
  bc.Encode(rc.encode_p1, tmp.data());
  rc.Stop();
  int len = (int)buf.GetBufPos();
  if (len > cap) return -len;
  std::memcpy(out, buf.GetBuf().data(), len);
  return len;
}

API int ref_bitplane_trace(const int32_t *s2u, int n, int maxbpn, uint16_t *p1s, uint8_t *bits,
                           int maxdec) {
  BitplaneCoder bc(maxbpn, n);
  std::vector<int32_t> tmp(s2u, s2u + n);
  int cnt = 0;
  EncodeP1 f = [&](uint32_t p1, int bit) {
    if (cnt < maxdec) { p1s[cnt] = (uint16_t)p1; bits[cnt] = (uint8_t)bit; }
    cnt++;
  };
  bc.Encode(f, tmp.data());
  return cnt;
}

API int ref_bitplane_decode(const uint8_t *in, int len, int n, int maxbpn, int32_t *err_out) {
  BufIO buf(len + 16);
  std::memcpy(buf.GetBuf().data(), in, len);
  buf.Reset();
  RangeCoderSH rc(buf, 1);
  rc.Init();
  BitplaneCoder bc(maxbpn, n);
  bc.Decode(rc.decode_p1, err_out);
  return 0;
}

// raw range coder: encode a given (p1,bit) sequence
API int ref_rangecoder_encode(const uint16_t *p1s, const uint8_t *bits, int n, uint8_t *out,
                              int cap) {
  BufIO buf;
  RangeCoderSH rc(buf);
  rc.Init();
  for (int i = 0; i < n; i++) rc.EncodeBitOne(p1s[i], bits[i]);
  rc.Stop();
  int len = (int)buf.GetBufPos();
  if (len > cap) return -len;
  std::memcpy(out, buf.GetBuf().data(), len);
  return len;
}

// ---------------------------------------------------------------- remap (sparse pcm)
// genuine Remap::Analyse + FrameCoder::CalcRemapError (libsac.cpp:230-251)
API double ref_remap(const int32_t *raw, int n, const int32_t *pred, const int32_t *error,
                     int32_t *s2u_map, int *maxbpn_map, uint8_t *usedl, uint8_t *usedh) {
  FrameCoder fc(1, n > 16 ? n : 16, make_cfg(4, 1, 1));
  fc.SetNumSamples(n);
  std::vector<int32_t> tmp(raw, raw + n);
  fc.framestats[0].mymap.Reset();
  fc.framestats[0].mymap.Analyse(tmp.data(), n);
  std::copy_n(pred, n, fc.pred[0].begin());
  std::copy_n(error, n, fc.error[0].begin());
  double r = fc.CalcRemapError(0, n);
  std::copy_n(fc.s2u_error_map[0].begin(), n, s2u_map);
  *maxbpn_map = fc.framestats[0].maxbpn_map;
  for (int i = 0; i <= (1 << 15); i++) {
    if (usedl) usedl[i] = fc.framestats[0].mymap.usedl[i];
    if (usedh) usedh[i] = fc.framestats[0].mymap.usedh[i];
  }
  return r;
}

// genuine MapEncoder over given used-flags, followed by nothing (range coder stopped)
API int ref_mapencode(const uint8_t *usedl, const uint8_t *usedh, uint8_t *out, int cap) {
  std::vector<bool> ul((1 << 15) + 1), uh((1 << 15) + 1);
  for (int i = 0; i <= (1 << 15); i++) { ul[i] = usedl[i]; uh[i] = usedh[i]; }
  BufIO buf;
  RangeCoderSH rc(buf);
  rc.Init();
  MapEncoder me(rc, ul, uh);
  me.Encode();
  rc.Stop();
  int len = (int)buf.GetBufPos();
  if (len > cap) return -len;
  std::memcpy(out, buf.GetBuf().data(), len);
  return len;
}

// ---------------------------------------------------------------- frame stats
API void ref_analyse(const int32_t *raw, int n, int32_t *out /*mean,min,max*/) {
  FrameCoder fc(1, n > 16 ? n : 16, make_cfg(4, 1, 1));
  std::copy_n(raw, n, fc.samples[0].begin());
  fc.AnalyseMonoChannel(0, n);
  out[0] = fc.framestats[0].mean;
  out[1] = fc.framestats[0].minval;
  out[2] = fc.framestats[0].maxval;
}

// ---------------------------------------------------------------- RNG / search helpers
// kinds: 0=r_01, 1=r_norm, 2=ru_int(0,arg)
API void ref_rng(int n, const int *kinds, const double *args, double *out) {
  Opt::box_const pb(1);
  pb[0] = {0, 1};
  OptProbe o(pb);
  for (int i = 0; i < n; i++) {
    if (kinds[i] == 0) out[i] = o.rand.r_01();
    else if (kinds[i] == 1) out[i] = o.rand.r_norm();
    else out[i] = (double)o.rand.ru_int(0, (uint32_t)args[i]);
  }
}

API void ref_gen_norm(double x, double xmin, double xmax, double r, int n, double *out) {
  Opt::box_const pb(1);
  pb[0] = {xmin, xmax};
  OptProbe o(pb);
  for (int i = 0; i < n; i++) out[i] = o.gen_norm(x, pb[0], r);
}

API double ref_reflect(double x, double xmin, double xmax) {
  Opt::box_const pb(1);
  pb[0] = {xmin, xmax};
  OptProbe o(pb);
  return o.reflect(x, xmin, xmax);
}

API void ref_ssc(int which, int n, const double *lambdas, double sigma0, double *out) {
  SSC0 s0(3, 50);
  SSC1 s1(0.05, 0.10, 0.05);
  double sigma = sigma0;
  for (int i = 0; i < n; i++) {
    sigma = which == 0 ? s0.update(sigma, lambdas[i]) : s1.update(sigma, lambdas[i]);
    out[i] = sigma;
  }
}

// DDS on an analytic test function f(x)=sum_i (i+1)*(x_i-c_i)^2 (search-loop parity)
API double ref_dds_quadratic(int ndim, const double *xmin, const double *xmax,
                             const double *xstart, const double *center, int nfunc_max,
                             int num_threads, double sigma, double *xbest, double *trace_cost) {
  Opt::box_const pb(ndim);
  vec1D xs(ndim);
  for (int i = 0; i < ndim; i++) { pb[i] = {xmin[i], xmax[i]}; xs[i] = xstart[i]; }
  OptDDS::DDSCfg c;
  c.nfunc_max = nfunc_max;
  c.num_threads = num_threads;
  c.sigma_init = sigma;
  DriverDDS dds(c, pb);
  std::vector<double> tc;
  dds.trace_cost = &tc;
  auto f = [&](const vec1D &x) {
    double s = 0;
    for (int i = 0; i < ndim; i++) { double d = x[i] - center[i]; s += (i + 1) * d * d; }
    return s;
  };
  auto ret = dds.run(f, xs);
  for (int i = 0; i < ndim; i++) xbest[i] = ret.second[i];
  if (trace_cost) for (size_t i = 0; i < tc.size() && (int)i < nfunc_max; i++) trace_cost[i] = tc[i];
  return ret.first;
}

// any searcher on the same analytic function: search 0 DDS, 1 DE, 2 CMA (settings as cmdline.cpp:221-241 derives them)
API double ref_search_quadratic(int search, int ndim, const double *xmin, const double *xmax, const double *xstart, const double *center,
                                int nfunc_max, int num_threads, double sigma, double *xbest, double *trace_cost, int *neval) {
  Opt::box_const pb(ndim);
  vec1D xs(ndim);
  for (int i = 0; i < ndim; i++) { pb[i] = {xmin[i], xmax[i]}; xs[i] = xstart[i]; }
  auto opt = make_searcher(search, nfunc_max, num_threads, sigma, pb);
  int ne = 0;
  auto f = [&](const vec1D &x) {
    double s = 0;
    for (int i = 0; i < ndim; i++) s += std::fabs(x[i] - center[i]) / (i + 1);   // no multiply-add: the same value with or without FMA contraction
    if (trace_cost) trace_cost[ne] = s;
    ne++;
    return s;
  };
  auto ret = opt->run(f, xs);
  for (int i = 0; i < ndim; i++) xbest[i] = ret.second[i];
  if (neval) *neval = ne;
  return ret.first;
}
API void ref_set_search_method(int search) { g_search_method = search; }
// 1: --opt-cfg=dds,N evaluates the N candidates of a generation on N threads (Opt::eval_points_mt); no traces in that mode
API void ref_set_parallel_eval(int on) { g_parallel_eval = on != 0; }

// ---------------------------------------------------------------- whole-frame encode/decode
struct ref_frame_cfg {
  int optimize;      // 0/1
  double fraction;   // search window fraction of max frame size
  int maxnfunc;      // E
  int num_threads;   // DDS N (0 = sequential run_single)
  double sigma;
  int optk;
  int cost;          // FrameCoder::SearchCost: 0 L1,1 RMS,2 Entropy,3 Golomb,4 Bitplane
  int reset;         // --opt-reset
  int sparse_pcm;
  int zero_mean;
};

// raw = planar [nch][n] un-centred samples.  profile_io: 58 floats in (warm start when
// reset==0) / out (profile written into the record).  Returns record length.
// trace_cost[maxnfunc], trace_coefs[maxnfunc*58] optional.
API int ref_encode_frame(int nch, int framesize, int n, const int32_t *raw,
                         const ref_frame_cfg *rc, float *profile_io, uint8_t *out, int cap,
                         double *trace_cost, float *trace_coefs, int *info /*per ch: maxbpn,mapped,size*/) {
  FrameCoder::tsac_cfg cfg = make_cfg(rc->optk, rc->sparse_pcm, rc->zero_mean);
  cfg.optimize = rc->optimize;
  cfg.ocfg.fraction = rc->fraction;
  cfg.ocfg.maxnfunc = rc->maxnfunc;
  cfg.ocfg.num_threads = rc->num_threads;
  cfg.ocfg.sigma = rc->sigma;
  cfg.ocfg.reset = rc->reset;
  cfg.ocfg.optimize_cost = (FrameCoder::SearchCost)rc->cost;
  cfg.ocfg.dds_cfg.nfunc_max = rc->maxnfunc;   // cmdline.cpp:222-226
  cfg.ocfg.dds_cfg.num_threads = rc->num_threads;
  cfg.ocfg.dds_cfg.sigma_init = rc->sigma;

  FrameCoder fc(nch, framesize, cfg);
  for (int ch = 0; ch < nch; ch++) std::copy_n(raw + (size_t)ch * n, n, fc.samples[ch].begin());
  fc.SetNumSamples(n);
  if (profile_io) set_profile(fc.base_profile, profile_io);

  // ---- RESTATED FrameCoder::Predict (libsac.cpp:443-479) around genuine members
  for (int ch = 0; ch < nch; ch++) {
    fc.AnalyseMonoChannel(ch, n);
    if (cfg.sparse_pcm) {
      fc.framestats[ch].mymap.Reset();
      fc.framestats[ch].mymap.Analyse(&(fc.samples[ch][0]), n);
    }
    if (cfg.zero_mean == 0) {
      fc.framestats[ch].mean = 0;
    } else if (fc.framestats[ch].mean != 0) {
      for (int i = 0; i < n; i++) fc.samples[ch][i] -= fc.framestats[ch].mean;
      fc.framestats[ch].minval -= fc.framestats[ch].mean;
      fc.framestats[ch].maxval -= fc.framestats[ch].mean;
    }
  }
  if (cfg.optimize) {
    if (cfg.ocfg.reset) fc.base_profile.LoadBaseProfile();
    std::vector<int> lp;
    for (int i = 0; i < (int)fc.base_profile.coefs.size(); i++)
      if (i != 56 && i != 57) lp.push_back(i);
    // ---- RESTATED FrameCoder::Optimize (libsac.cpp:365-427)
    SacProfile &profile = fc.base_profile;
    int samples_to_optimize = std::min(n, static_cast<int>(std::ceil(framesize * cfg.ocfg.fraction)));
    const int start_pos = (n - samples_to_optimize) / 2;
    CostFunction *CostFunc = nullptr;
    switch (cfg.ocfg.optimize_cost) {
      case FrameCoder::SearchCost::L1: CostFunc = new CostL1(); break;
      case FrameCoder::SearchCost::RMS: CostFunc = new CostRMS(); break;
      case FrameCoder::SearchCost::Golomb: CostFunc = new CostGolomb(); break;
      case FrameCoder::SearchCost::Entropy: CostFunc = new CostEntropy(); break;
      case FrameCoder::SearchCost::Bitplane: CostFunc = new CostBitplane(); break;
    }
    const int ndim = lp.size();
    vec1D xstart(ndim);
    Opt::box_const pb(ndim);
    for (int i = 0; i < ndim; i++) {
      pb[i].xmin = profile.coefs[lp[i]].vmin;
      pb[i].xmax = profile.coefs[lp[i]].vmax;
      xstart[i] = profile.coefs[lp[i]].vdef;
    }
    std::atomic<int> neval{0};
    auto cost_func = [&](const vec1D &x) {
      FrameCoder::tch_samples tmp_error(nch, std::vector<int32_t>(samples_to_optimize));
      SacProfile tmp_profile = profile;
      for (int i = 0; i < ndim; i++) tmp_profile.coefs[lp[i]].vdef = x[i];
      fc.PredictFrame(tmp_profile, tmp_error, start_pos, samples_to_optimize, true);
      double c = fc.GetCost(CostFunc, tmp_error, samples_to_optimize);
      if (!g_parallel_eval && neval < rc->maxnfunc + 32) {   // + 32: DE evaluates its whole start-up population (30 points) even when maxnfunc is smaller
        if (trace_cost) trace_cost[neval] = c;
        if (trace_coefs)
          for (int i = 0; i < 58; i++) trace_coefs[(size_t)neval * 58 + i] = tmp_profile.coefs[i].vdef;
      }
      neval++;
      return c;
    };
    std::unique_ptr<Opt> searcher = make_searcher(g_search_method, rc->maxnfunc, rc->num_threads, rc->sigma, pb);
    Opt::ppoint ret = searcher->run(cost_func, xstart);
    for (int i = 0; i < ndim; i++) profile.coefs[lp[i]].vdef = ret.second[i];
    delete CostFunc;
  }
  fc.PredictFrame(fc.base_profile, fc.error, 0, n, false);
  fc.CnvError_S2U(fc.error, n);
  // ---- genuine Encode + WriteEncoded
  fc.Encode();
  std::string fn = tmpname();
  int len = 0;
  {
    AudioFile f;
    if (f.OpenWrite(fn) != 0) return -1;
    fc.WriteEncoded(f);
    f.Close();
  }
  FILE *fp = fopen(fn.c_str(), "rb");
  if (!fp) return -1;
  fseek(fp, 0, SEEK_END);
  len = (int)ftell(fp);
  fseek(fp, 0, SEEK_SET);
  int ret = len;
  if (len <= cap) { if (fread(out, 1, len, fp) != (size_t)len) ret = -1; }
  else ret = -len;
  fclose(fp);
  unlink(fn.c_str());
  if (profile_io)
    for (int i = 0; i < 58; i++) profile_io[i] = fc.base_profile.coefs[i].vdef;
  if (info)
    for (int ch = 0; ch < nch; ch++) {
      info[3 * ch + 0] = fc.framestats[ch].enc_mapped ? fc.framestats[ch].maxbpn_map : fc.framestats[ch].maxbpn;
      info[3 * ch + 1] = fc.framestats[ch].enc_mapped;
      info[3 * ch + 2] = fc.framestats[ch].blocksize;
    }
  return ret;
}

// genuine ReadEncoded + Decode + Unpredict.  out = planar [nch][n_out].
API int ref_decode_frame(const uint8_t *rec, int len, int nch, int framesize, int32_t *out,
                         int cap_samples, float *coefs_out) {
  std::string fn = tmpname();
  FILE *fp = fopen(fn.c_str(), "wb");
  if (!fp) return -1;
  fwrite(rec, 1, len, fp);
  fclose(fp);
  FrameCoder::tsac_cfg cfg = make_cfg(4, 1, 1);
  FrameCoder fc(nch, framesize, cfg);
  int n = -1;
  {
    AudioFile f;
    if (f.OpenRead(fn) != 0) { unlink(fn.c_str()); return -1; }
    fc.ReadEncoded(f);
    f.Close();
  }
  unlink(fn.c_str());
  n = fc.GetNumSamples();
  if (n > cap_samples || n > framesize) return -2;
  fc.Decode();
  fc.Unpredict();
  for (int ch = 0; ch < nch; ch++) std::copy_n(fc.samples[ch].begin(), n, out + (size_t)ch * n);
  if (coefs_out)
    for (int i = 0; i < 58; i++) coefs_out[i] = fc.base_profile.coefs[i].vdef;
  return n;
}



// ---------------------------------------------------------------- .sac container (genuine reader)
// Opens a .sac file with the genuine Sac class (file/sac.cpp): ReadSACHeader, ReadMD5, then the frame
// loop of Codec::DecodeFile (libsac.cpp:856-880) with the genuine FrameCoder.  hdr = {numchannels,
// samplerate, bitspersample, numsamples, max_framelen, metadatasize}; pcm_out planar [nch][numsamples].
API int ref_read_sac(const char *path, int *hdr, uint8_t *md5, uint8_t *meta, int metacap, int32_t *pcm_out, long long cap) {
  Sac sac;
  if (sac.OpenRead(path) != 0) return -1;
  if (sac.ReadSACHeader() != 0) { sac.Close(); return -2; }
  sac.ReadMD5(md5);
  hdr[0] = sac.getNumChannels(); hdr[1] = sac.getSampleRate(); hdr[2] = sac.getBitsPerSample(); hdr[3] = sac.getNumSamples();
  hdr[4] = sac.mcfg.max_framelen; hdr[5] = (int)sac.mcfg.metadatasize;
  if ((int)sac.metadata.size() <= metacap) std::copy(sac.metadata.begin(), sac.metadata.end(), meta);
  const int nch = hdr[0], total = hdr[3];
  if ((long long)nch * total > cap) { sac.Close(); return -3; }
  FrameCoder::tsac_cfg cfg = make_cfg(4, 1, 1);
  FrameCoder fc(nch, (int)sac.mcfg.max_framesize, cfg);
  int done = 0, frames = 0;
  while (done < total) {
    fc.ReadEncoded(sac);
    fc.Decode();
    fc.Unpredict();
    const int n = fc.GetNumSamples();
    if (n <= 0 || done + n > total) { sac.Close(); return -4; }
    for (int ch = 0; ch < nch; ch++) std::copy_n(fc.samples[ch].begin(), n, pcm_out + (size_t)ch * total + done);
    done += n; frames++;
  }
  sac.Close();
  return frames;
}

// ---------------------------------------------------------------- adaptive sub-frame split
// Genuine Codec::Analyse (+ AnalyseSparse / PushState, SparsePCM) on one read of `samples_read`
// samples per channel.  out: triples {start, length, state}.
API int ref_plan_subframes(int nch, int samples_read, const int32_t *pcm, long long ch_stride, int blocksamples,
                           int min_frame_length, int *out, int cap) {
  FrameCoder::tsac_cfg cfg = make_cfg(4, 1, 1);
  cfg.verbose_level = 0;
  Codec codec(cfg);
  std::vector<std::vector<int32_t>> samples(nch, std::vector<int32_t>(samples_read));
  for (int ch = 0; ch < nch; ch++) std::copy_n(pcm + (size_t)ch * ch_stride, samples_read, samples[ch].begin());
  auto sf = codec.Analyse(samples, blocksamples, min_frame_length, samples_read);
  int n = 0;
  for (auto &f : sf) { if (n < cap) { out[3 * n] = f.start; out[3 * n + 1] = f.length; out[3 * n + 2] = f.state; } n++; }
  return n;
}
// SparsePCM::Analyse on one buffer -> {fraction_used, fraction_cost}
API void ref_sparse_cost(const int32_t *buf, int n, double *out2) {
  SparsePCM sp;
  sp.Analyse(std::span<const int32_t>(buf, (size_t)n));
  out2[0] = sp.fraction_used; out2[1] = sp.fraction_cost;
}

API int ref_abi_version() { return 1; }

// ---------------------------------------------------------------- math micro-probes
#include "common/math.h"
#include "common/alignbuf.h"
API double ref_dot(const double *x, const double *y, int n) {
  return slmath::dot(std::span<const double>(x, (size_t)n), std::span<const double>(y, (size_t)n));
}
API double ref_s2pow(const double *x, const double *p, int n) {
  std::vector<double, align_alloc<double>> pa(p, p + n);
  return slmath::calc_s2pow(std::span<const double>(x, (size_t)n), pa);
}
API int ref_ldlt(const double *A, int n, double nu, const double *b, double *w) {
  vec2D m(n, vec1D(n));
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) m[i][j] = A[i * n + j];
  slmath::LDLT ldlt(n);
  vec1D bb(b, b + n), ww(w, w + n);
  int ok = ldlt.Factor(m, nu);
  if (ok) ldlt.Solve(bb, ww);
  for (int i = 0; i < n; i++) w[i] = ww[i];
  return ok;
}
