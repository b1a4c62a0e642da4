// tests/emu/emu.cpp -- CPU execution of the *product's kernel bodies* (sac_amd/csrc/pred_*.h,
// coder.h) through the lane-serial ExecEmu executor.  Test infrastructure only: lets the
// `-m "not gpu"` suite check the kernels' logic (right-looking LDLT, fused NLMS sweep, ring
// indexing, stereo geometry ...) against the oracle without a GPU.  The product never builds
// or loads this file.
#include <random>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include "../../sac_amd/csrc/pred_ols.h"
#include "../../sac_amd/csrc/pred_ols_pack.h"
#include "../../sac_amd/csrc/pred_ols_grid.h"
#include "../../sac_amd/csrc/pred_lms.h"
#include "../../sac_amd/csrc/pred_bias.h"
#include "../../sac_amd/csrc/pred_tables.h"

using namespace sacamd;
#define API extern "C" __attribute__((visibility("default")))

template <class C, int NL = 256, int CANON = 0, int ROUNDS = 1>
static void run_lms(const ChanParam &p, const double *sp, const double *tab, const int *self, int n, double *pio, const double *tabc = nullptr) {
  const int rc0[4] = {p.vn[0] + 1, p.vn[1] + 1, p.vn[2] + 1, p.vn[3] + 1};
  std::vector<char> lds(LmsLds<NL, C, CANON>::bytes(rc0), (char)0xFF);   // tight, as the launcher sizes the dynamic LDS; LDS is not zeroed on the device: start from NaN bit patterns
  ExecEmu<NL> *ex = new ExecEmu<NL>;
  const int rc[4] = {p.vn[0] + 1, p.vn[1] + 1, p.vn[2] + 1, p.vn[3] + 1};   // tight rings, as the host sizes them
  lms_stage<ExecEmu<NL>, C, CANON, ROUNDS>(*ex, p, sp, tab, self, n, pio, pio, lds.data(), rc, nullptr, nullptr, tabc);
  delete ex;
}

static int g_ols_grid = 1;      // 17/25..64 taps: 1 = grid kernel (the launcher's choice since round 5), 0 = the row-per-lane / panel kernels
API void emu_set_ols_grid(int on) { g_ols_grid = on; }

// samples planar [nch][total] mean-removed; stats [nch][3] = {min,max,mean}
API int emu_predict(int nch, int total, const int32_t *samples, const int32_t *stats, const float *coefs,
                    int from, int n, int optimize, int optk, double *plpc, double *psum,
                    int32_t *err, int32_t *pred) {
  FrameStatsD st[2] = {};
  for (int ch = 0; ch < nch; ch++) { st[ch].minval = stats[3 * ch]; st[ch].maxval = stats[3 * ch + 1]; st[ch].mean = stats[3 * ch + 2]; st[ch].numsamples = total; }
  ChanParam cp[2]; int ch_ref = 0;
  map_profile(coefs, optimize != 0, optk, nch, st, cp, &ch_ref);
  for (int slot = 0; slot < nch; slot++) {
    const int ch_self = (nch == 2) ? (slot == 0 ? ch_ref : 1 - ch_ref) : 0;
    const int ch_other = (nch == 2) ? 1 - ch_self : 0;
    const ChanParam &p = cp[slot];
    const int32_t *self = samples + (size_t)ch_self * total + from;
    const int32_t *other = samples + (size_t)ch_other * total + from;
    double *pl = plpc + (size_t)ch_self * n, *ps = psum + (size_t)ch_self * n;
    if (p.n_ols > kMaxOLS) return -1;
    {
#define EMU_OLSP(NM) { std::vector<char> lds(ols_panel_lds_bytes(NM, 4)); ExecEmu<256> ex; ols_stage_panel<ExecEmu<256>, NM>(ex, p, self, other, n, pl, lds.data()); }
#define EMU_OLSG(NBK) { std::vector<char> lds(OlsLdsGrid::bytes(8 * NBK), (char)0xFF); ExecEmu<64> ex; ols_stage_grid<ExecEmu<64>, NBK>(ex, p, self, other, n, pl, lds.data()); }
#define EMU_OLS(NM) { std::vector<char> lds(OlsLdsFast::bytes(NM)); ExecEmu<64> ex; ols_stage_reg<ExecEmu<64>, NM>(ex, p, self, other, n, pl, lds.data()); }
      if (g_ols_grid && p.n_ols > (optimize ? 24 : 16) && p.n_ols <= 64) {     // 2D-cyclic one-wave kernel (pred_ols_grid.h), as the launcher: 25..64 taps, and 17..24 in the final pass
        if (p.n_ols <= 24) EMU_OLSG(3) else if (p.n_ols <= 32) EMU_OLSG(4) else if (p.n_ols <= 40) EMU_OLSG(5) else if (p.n_ols <= 48) EMU_OLSG(6) else if (p.n_ols <= 56) EMU_OLSG(7) else EMU_OLSG(8)
      }
      else if (!optimize) {     // as the host (build_items): the final pass uses the 32 / 64 / 96-tap capacity classes only
        if (p.n_ols <= 32) EMU_OLS(32)
        else if (p.n_ols <= 56) EMU_OLSP(56)
        else if (p.n_ols <= 64) EMU_OLSP(64)
        else { std::vector<char> lds(ols_panel2_lds_bytes(96)); ExecEmu<256> ex; ols_stage_panel2<ExecEmu<256>, 96>(ex, p, self, other, n, pl, lds.data()); }
      }
      else if (p.n_ols <= 16) EMU_OLS(16)
      else if (p.n_ols <= 24) EMU_OLS(24)
      else if (p.n_ols <= 32) EMU_OLS(32)
      else if (p.n_ols <= 40) EMU_OLS(40)     // as the launcher: one wave up to 64 taps
      else if (p.n_ols <= 48) EMU_OLS(48)
      else if (p.n_ols <= 56) EMU_OLS(56)
      else if (p.n_ols <= 64) EMU_OLS(64)
      else { std::vector<char> lds(ols_panel2_lds_bytes(96)); ExecEmu<256> ex; ols_stage_panel2<ExecEmu<256>, 96>(ex, p, self, other, n, pl, lds.data()); }   // 65..96 taps: two rows per lane
    }
    std::vector<double> tab; double sp[4];
    for (int s = 0; s < 4; s++) {
      size_t o = tab.size(); tab.resize(o + 2 * (size_t)p.vn[s]);
      double sum = 0;
      for (int i = 0; i < p.vn[s]; i++) { lms_table_entry(i, p.vmudecay[s], p.vpowdecay[s], &tab[o + i], &tab[o + p.vn[s] + i]); sum += tab[o + p.vn[s] + i]; }
      sp[s] = sum;
    }
    for (int t = 0; t < n; t++) ps[t] = pl[t];
    const int *vn = p.vn;
    const bool systolic = std::getenv("SACAMD_CANON_SYSTOLIC") && std::getenv("SACAMD_CANON_SYSTOLIC")[0] == '1';
    const int c3 = canon3_class_for(vn);
    if (!optimize && !systolic && c3 == 10) run_lms<LmsClass<17, 0, 0, 0>, 256, 3>(p, sp, tab.data(), self, n, ps);       // as the launcher (lms_class_for): lane-map canonical layouts
    else if (!optimize && !systolic && c3 == 11) run_lms<LmsClass<33, 0, 0, 0>, 256, 3>(p, sp, tab.data(), self, n, ps);
    else if (!optimize && !systolic && c3 == 12) run_lms<LmsClass<49, 0, 0, 0>, 256, 3>(p, sp, tab.data(), self, n, ps);
    else if (!optimize && !systolic && c3 == 13) run_lms<LmsClass<33, 0, 0, 0>, 512, 3>(p, sp, tab.data(), self, n, ps);
    else if (!optimize) {   // the systolic layouts of round 2: fallback for the largest profiles; the final pass sums in slmath::dot order
      // lane-major table copies as k_tables writes them for the canonical layouts (pred_tables.h)
      const int rounds = (vn[0] <= 2304 && vn[1] <= 1280 && vn[2] <= 768 && vn[3] <= 256) ? 1 : ((vn[0] <= 4608 && vn[1] <= 2560 && vn[2] <= 1536 && vn[3] <= 512) ? 2 : 4);
      std::vector<double> tabc((size_t)canon_tab_doubles(rounds), std::nan(""));
      {
        size_t on = 0, oc = 0;
        for (int s = 0; s < 4; s++) {
          const int J = canon_slots(s), n8 = vn[s] >= 8 ? vn[s] & ~7 : 0, n4 = vn[s] >= 8 ? vn[s] & ~3 : 0;
          double *mtc = &tabc[oc], *ptc = mtc + (size_t)rounds * J * kCanonNL;
          for (int i = 0; i < n8; i++) mtc[canon_mt_index(J, i)] = tab[on + i];
          for (int i = 0; i < n4; i++) ptc[canon_pt_index(J, i)] = tab[on + vn[s] + i];
          on += 2 * (size_t)vn[s]; oc += (size_t)canon_stage_doubles(s, rounds);
        }
      }
      if (rounds == 1) run_lms<LmsClass<9, 5, 3, 1>, 256, 2, 1>(p, sp, tab.data(), self, n, ps, tabc.data());
      else if (rounds == 2) run_lms<LmsClass<9, 5, 3, 1>, 256, 2, 2>(p, sp, tab.data(), self, n, ps, tabc.data());
      else run_lms<LmsClass<9, 5, 3, 1>, 256, 2, 4>(p, sp, tab.data(), self, n, ps, tabc.data());
    }
    else if (vn[0] <= 2048 && vn[1] <= 1024 && vn[2] <= 512 && vn[3] <= 256) run_lms<LmsClass<8, 4, 2, 1>>(p, sp, tab.data(), self, n, ps);
    else if (vn[0] <= 1536 && vn[1] <= 2560 && vn[2] <= 1024 && vn[3] <= 512) run_lms<LmsClass<6, 10, 4, 2>>(p, sp, tab.data(), self, n, ps);
    else if (vn[0] <= 3328 && vn[1] <= 1280 && vn[2] <= 768 && vn[3] <= 256) run_lms<LmsClass<13, 5, 3, 1>>(p, sp, tab.data(), self, n, ps);
    else if (vn[0] <= 3584 && vn[1] <= 512 && vn[2] <= 1024 && vn[3] <= 512) run_lms<LmsClass<14, 2, 4, 2>>(p, sp, tab.data(), self, n, ps);     // 14, 15: round-6 layouts
    else if (vn[0] <= 2304 && vn[1] <= 1280 && vn[2] <= 1280 && vn[3] <= 768) run_lms<LmsClass<9, 5, 5, 3>>(p, sp, tab.data(), self, n, ps);
    else if (vn[0] <= 3072 && vn[1] <= 3072 && vn[2] <= 1024 && vn[3] <= 512) run_lms<LmsClass<12, 12, 4, 2>>(p, sp, tab.data(), self, n, ps);
    else if (vn[0] <= 4608 && vn[1] <= 1536 && vn[2] <= 1024 && vn[3] <= 512) run_lms<LmsClass<18, 6, 4, 2>>(p, sp, tab.data(), self, n, ps);
    else if (vn[0] <= 3840 && vn[1] <= 2048 && vn[2] <= 1536 && vn[3] <= 256) run_lms<LmsClass<15, 8, 6, 1>>(p, sp, tab.data(), self, n, ps);
    else run_lms<LmsClass<16, 8, 4, 2>, 512>(p, sp, tab.data(), self, n, ps);   // as the launcher: 512 lanes
    std::vector<double> tables(kBiasSlabDoubles);
    bias_stage(p, self, n, ps, stats[3 * ch_self + 2], err + (size_t)ch_self * n, pred ? pred + (size_t)ch_self * n : nullptr, tables.data());
  }
  return 0;
}

// Packed OLS stage (pred_ols_pack.h): `count` mono/stereo-slot work-items, G = 64 / GL per emulated wave, in list order.
// Item i: samples_i planar [nch_i][total_i] (mean-removed), stats_i [nch_i][3], coefs_i [58], slot_i (0 / 1), n_i samples from 0;
// output plpc_i [n_i].  cls: 0 = <16, 16 lanes>, 1 = <24, 32 lanes>, 2 = <32, 32 lanes>.  Returns 0, or -1 when an item does not fit.
API int emu_ols_pack(int cls, int count, const int *nch, const int *total, const int32_t *const *samples, const int32_t *const *stats,
                     const float *const *coefs, const int *slot, const int *n, int optimize, int optk, double *const *plpc) {
  std::vector<ChanParam> cps(count);
  std::vector<const int *> selfs(count), others(count);
  const int nmaxc = cls == 0 ? 16 : cls == 1 ? 24 : 32;
  for (int i = 0; i < count; i++) {
    FrameStatsD st[2] = {};
    for (int ch = 0; ch < nch[i]; ch++) { st[ch].minval = stats[i][3 * ch]; st[ch].maxval = stats[i][3 * ch + 1]; st[ch].mean = stats[i][3 * ch + 2]; st[ch].numsamples = total[i]; }
    ChanParam cp[2]; int ch_ref = 0;
    map_profile(coefs[i], optimize != 0, optk, nch[i], st, cp, &ch_ref);
    const int ch_self = (nch[i] == 2) ? (slot[i] == 0 ? ch_ref : 1 - ch_ref) : 0;
    const int ch_other = (nch[i] == 2) ? 1 - ch_self : 0;
    cps[i] = cp[slot[i]];
    if (cps[i].n_ols > nmaxc) return -1;
    selfs[i] = samples[i] + (size_t)ch_self * total[i];
    others[i] = samples[i] + (size_t)ch_other * total[i];
  }
  const int GL = cls == 0 ? 16 : 32, G = 64 / GL;
  for (int b = 0; b < count; b += G) {
    ExecEmu<64> ex;
    ExecEmu<64>::Reg<OlsPackSlot> sl;
    for (int l = 0; l < 64; l++) {
      const int i = b + l / GL;
      if (i < count) sl[l] = OlsPackSlot{&cps[i], selfs[i], others[i], plpc[i], n[i]};
      else sl[l] = OlsPackSlot{&cps[b], selfs[b], others[b], nullptr, 0};
    }
    const int kk = cps[b].k;
    if (cls == 0) { std::vector<char> lds(ols_pack_lds_bytes<16, 16>(), (char)0xFF); ols_stage_pack<ExecEmu<64>, 16, 16>(ex, sl, kk, lds.data()); }
    else if (cls == 1) { std::vector<char> lds(ols_pack_lds_bytes<24, 32>(), (char)0xFF); ols_stage_pack<ExecEmu<64>, 24, 32>(ex, sl, kk, lds.data()); }
    else { std::vector<char> lds(ols_pack_lds_bytes<32, 32>(), (char)0xFF); ols_stage_pack<ExecEmu<64>, 32, 32>(ex, sl, kk, lds.data()); }
  }
  return 0;
}

// cascade layout class the launcher picks for the final pass of an item with these stage lengths (kernels_pred.hip,
// lms_class_for): 10..13 = lane-map layouts (J, lanes) = (17,256) (33,256) (49,256) (33,512); 9 = systolic fallback
API int emu_canon_class(const int *vn) {
  const int c3 = canon3_class_for(vn);
  return c3 >= 0 ? c3 : 9;
}

// LDS bytes the lane-map layout `cls` (10..13) asks for when it holds an item with stage lengths vn (the launcher's dynamic LDS size)
API long emu_canon_lds_bytes(const int *vn, int cls) {
  const int rc[4] = {vn[0] + 1, vn[1] + 1, vn[2] + 1, vn[3] + 1};
  switch (cls) {
    case 10: return (long)LmsLds<256, LmsClass<17, 0, 0, 0>, 3>::bytes(rc);
    case 11: return (long)LmsLds<256, LmsClass<33, 0, 0, 0>, 3>::bytes(rc);
    case 12: return (long)LmsLds<256, LmsClass<49, 0, 0, 0>, 3>::bytes(rc);
    case 13: return (long)LmsLds<512, LmsClass<33, 0, 0, 0>, 3>::bytes(rc);
    default: return -1;
  }
}

// ---------------------------------------------------------------- coder
#include "../../sac_amd/csrc/coder.h"
#include <cmath>
static void host_laplace(std::vector<unsigned short> &lap, unsigned short *plap) {
  lap.resize((size_t)kLaplacePlanes * kLaplaceAvg);
  for (int b = 0; b < kLaplacePlanes; b++)
    for (int a = 0; a < kLaplaceAvg; a++) {
      double p_l = 0.0;
      if (a > 0) { double theta = std::exp(-1.0 / a); p_l = 1.0 - 1.0 / (1 + std::pow(theta, (double)(1 << b))); }
      int p1 = std::min(std::max((int)std::round(p_l * kPScale), 1), (int)kPScaleM);
      lap[(size_t)b * kLaplaceAvg + a] = (unsigned short)p1;
    }
  for (int i = 0; i < 32; i++) {
    double pw = std::pow(0.99, (double)(i < 31 ? (1 << i) : -2147483647 - 1));
    plap[i] = (unsigned short)std::min(std::max((int)std::round((1.0 - 1.0 / (1 + pw)) * kPScale), 1), (int)kPScaleM);
  }
}
API int emu_bitplane(const int32_t *s2u, int n, int maxbpn, const unsigned char *used, const int *fwd_i, const int *inv_i, unsigned char *out, int cap) {
  static std::vector<unsigned short> lap; static unsigned short plap[32];
  if (lap.empty()) host_laplace(lap, plap);
  std::vector<short> gf(kPScale); std::vector<unsigned short> gi(4095);
  for (int i = 0; i < kPScale; i++) gf[i] = (short)fwd_i[i];
  for (int i = 0; i < 4095; i++) gi[i] = (unsigned short)inv_i[i];
  std::vector<CntL> csig0(65536);
  CoderModel *M = new CoderModel; CoderTabs *T = new CoderTabs; CoderWin *W = new CoderWin; MapModel *MM = new MapModel;
  ExecEmu<64> ex;
  ex.par([&](int l) { coder_tabs_init(*T, gf.data(), gi.data(), l, 64); });
  int len = coder_stream(ex, s2u, n, maxbpn, used, lap.data(), gf.data(), gi.data(), plap, csig0.data(), out, cap, *M, *T, *W, *MM);
  delete M; delete T; delete W; delete MM;
  return len;
}

// laplace_direct (coder.h: PredictLaplace evaluated in the kernel for wide material) vs the host libm expression of the reference
API long emu_laplace_mismatches(long n, int seed) {
  std::mt19937_64 g(seed);
  long bad = 0;
  for (long i = 0; i < n; i++) {
    const int b = (int)(g() % 27);
    const int ea = (int)(g() % 27);                       // avg spread over all magnitudes up to 2^27
    const unsigned a = (unsigned)((g() % (1ull << ea)) + (i & 1 ? (1ull << ea) : 0));
    double p_l = 0.0;
    if (a > 0) { double theta = std::exp(-1.0 / a); p_l = 1.0 - 1.0 / (1 + std::pow(theta, (double)(1 << b))); }
    const int want = std::min(std::max((int)std::round(p_l * kPScale), 1), (int)kPScaleM);
    if (laplace_direct(a, b) != want) bad++;
  }
  return bad;
}

// canon.h: the double -> int32 conversion of libsac.cpp:106 as the reference's x86-64 build executes it
API int emu_cvt_i32(double r) { return cvt_i32_x86(r); }

// decode side: bytes -> s2u values (+ used flags when with_map); returns bytes consumed
API int emu_bitplane_decode(const unsigned char *in, int inlen, int n, int maxbpn, unsigned char *used_out, const int *fwd_i, const int *inv_i, int32_t *s2u) {
  static std::vector<unsigned short> lap; static unsigned short plap[32];
  if (lap.empty()) host_laplace(lap, plap);
  std::vector<short> gf(kPScale); std::vector<unsigned short> gi(4095);
  for (int i = 0; i < kPScale; i++) gf[i] = (short)fwd_i[i];
  for (int i = 0; i < 4095; i++) gi[i] = (unsigned short)inv_i[i];
  std::vector<CntL> csig0(65536);
  CoderModel *M = new CoderModel; CoderTabs *T = new CoderTabs; CoderWin *W = new CoderWin; MapModel *MM = new MapModel;
  ExecEmu<64> ex;
  ex.par([&](int l) { coder_tabs_init(*T, gf.data(), gi.data(), l, 64); });
  const int used = coder_stream_dec(ex, in, inlen, n, maxbpn, used_out, lap.data(), plap, csig0.data(), s2u, *M, *T, *W, *MM);
  delete M; delete T; delete W; delete MM;
  return used;
}

// ---------------------------------------------------------------- host DDS logic (product code) on a test function
#include "../../sac_amd/csrc/dds_host.h"
API double emu_dds_quadratic(int ndim, const double *xmin, const double *xmax, const double *xstart, const double *center,
                             int nfunc_max, int num_threads, double sigma, double *xbest, double *trace_cost) {
  std::vector<Coef> box(ndim); std::vector<int> lp(ndim);
  for (int i = 0; i < ndim; i++) { box[i] = {(float)xmin[i], (float)xmax[i], 0.f}; lp[i] = i; }
  int ne = 0;
  auto f = [&](const std::vector<double> &x) {
    double s = 0;
    for (int i = 0; i < ndim; i++) { double d = x[i] - center[i]; s += (i + 1) * d * d; }
    if (trace_cost && ne < nfunc_max) trace_cost[ne] = s;
    ne++;
    return s;
  };
  FrameSearch fs; fs.sigma = sigma; fs.xb.assign(xstart, xstart + ndim); fs.cb = f(fs.xb);
  while (fs.nfunc < nfunc_max) {
    const int nt = num_threads <= 0 ? 1 : std::min(nfunc_max - fs.nfunc, num_threads);
    fs.gen.clear();
    for (int i = 0; i < nt; i++) { fs.gen.push_back(fs.candidate(box.data(), lp, nfunc_max)); fs.nfunc++; }
    std::vector<double> gc(nt);
    for (int i = 0; i < nt; i++) gc[i] = f(fs.gen[i]);
    if (num_threads <= 0) fs.select_single(gc[0]); else fs.select_mt(gc.data(), nt);
  }
  for (int i = 0; i < ndim; i++) xbest[i] = fs.xb[i];
  return fs.cb;
}

// the DE / CMA searchers (product code, search_host.h) on the same test function: search 1 = DE, 2 = CMA
#include "../../sac_amd/csrc/search_host.h"
template <class S>
static double run_searcher_quadratic(S &sr, int ndim, const double *xmin, const double *xmax, const double *xstart, const double *center,
                                     int nfunc_max, double sigma, double *xbest, double *trace_cost, int *neval) {
  SearchBox box; box.lo.assign(xmin, xmin + ndim); box.hi.assign(xmax, xmax + ndim);
  int ne = 0;
  auto f = [&](const std::vector<double> &x) {
    double s = 0;
    for (int i = 0; i < ndim; i++) s += std::fabs(x[i] - center[i]) / (i + 1);   // no multiply-add: the same value with or without FMA contraction
    if (trace_cost) trace_cost[ne] = s;
    ne++;
    return s;
  };
  std::vector<double> x0(xstart, xstart + ndim);
  sr.start(&box, x0, f(x0), nfunc_max, sigma);
  for (;;) {
    const auto &gen = sr.propose();
    if (gen.empty()) break;
    std::vector<double> gc(gen.size());
    for (size_t i = 0; i < gen.size(); i++) gc[i] = f(gen[i]);
    sr.accept(gc.data());
  }
  if (neval) *neval = ne;
  return 0.0;
}
API double emu_search_quadratic(int search, int ndim, const double *xmin, const double *xmax, const double *xstart, const double *center,
                                int nfunc_max, int num_threads, double sigma, double *xbest, double *trace_cost, int *neval) {
  (void)num_threads;
  if (search == 1) {
    FrameSearchDE de;
    run_searcher_quadratic(de, ndim, xmin, xmax, xstart, center, nfunc_max, sigma, xbest, trace_cost, neval);
    for (int i = 0; i < ndim; i++) xbest[i] = de.best.x[i];
    return de.best.cost;
  }
  FrameSearchCMA cma;
  run_searcher_quadratic(cma, ndim, xmin, xmax, xstart, center, nfunc_max, sigma, xbest, trace_cost, neval);
  for (int i = 0; i < ndim; i++) xbest[i] = cma.xbest[i];
  return cma.cbest;
}

// ---------------------------------------------------------------- libm port vs the host libm
#include <random>
API long emu_libm_mismatches(long n, int seed) {
  std::mt19937_64 g(seed); std::uniform_real_distribution<double> U(0, 1);
  long bad = 0;
  for (long i = 0; i < n; i++) {
    double x = (i & 1) ? -U(g) * 760 : (U(g) - 0.5) * 100;
    if ((i % 7) == 0) x = -U(g) * 1e-3;
    volatile double a = std::exp(x); if (sa_asu(a) != sa_asu(sa_exp(x))) bad++;
    double px, py; const int m = i % 4;
    if (m == 0) { px = 0.1 + U(g) * 1e5; py = -(0.1 + U(g) * 1.9); } else if (m == 1) { px = 1 + (double)(g() % 8192); py = U(g); }
    else if (m == 2) { px = 0.98 + U(g) * 0.02; py = (double)(g() % 8192); } else { px = std::exp((U(g) - 0.5) * 20); py = (U(g) - 0.5) * 40; }
    volatile double b = std::pow(px, py); if (sa_asu(b) != sa_asu(sa_pow(px, py))) bad++;
  }
  return bad;
}
