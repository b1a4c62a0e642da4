// sac_amd/csrc/kernels_pred.hip -- gfx950 kernels of the three predictor stages.
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (explicit fma only; see canon.h).
#include <algorithm>
#include "kernels.h"
#include "pred_bias.h"
#include "pred_lms.h"
#include "pred_ols.h"
#include "pred_ols_pack.h"
#include "pred_ols_grid.h"
#include "pred_tables.h"

#include <atomic>
#include <cstdlib>

namespace sacamd {

// Opt a kernel in to more than 64 KB of dynamic LDS once per DEVICE (the attribute is per device; contexts on
// different devices, and launchers called from several host threads, share these functions).  Idempotent, so a
// race between two first launches is harmless.
static hipError_t ensure_dyn_lds(const void *fn, size_t bytes, std::atomic<unsigned long long> &done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// ------------------------------------------------------------------ NLMS tables
// grid (work-item, stage); mutab/powtab per tap, then the in-order sum of powtab (ls.h:37-42)
__global__ __launch_bounds__(256) void k_tables(WorkItem *items, double *tab) {
  __shared__ double chunk[256];
  WorkItem &it = items[blockIdx.x];
  const int s = blockIdx.y;
  long long off = it.off_tab;
  for (int q = 0; q < s; q++) off += 2LL * it.p.vn[q];
  const int ns = it.p.vn[s];
  double *mt = tab + off, *pt = tab + off + ns;
  const double mud = it.p.vmudecay[s], pwd = it.p.vpowdecay[s];
  double *mtc = nullptr, *ptc = nullptr;        // lane-major copies for the canonical-order layouts (final pass)
  int J = 1, n8 = 0, n4 = 0;
  if (it.off_tabc >= 0) {
    const int rounds = canon_rounds_of_class(it.lms_class);
    long long oc = it.off_tabc;
    for (int q = 0; q < s; q++) oc += canon_stage_doubles(q, rounds);
    J = canon_slots(s);
    mtc = tab + oc; ptc = mtc + (long long)rounds * J * kCanonNL;
    n8 = ns >= 8 ? ns & ~7 : 0; n4 = ns >= 8 ? ns & ~3 : 0;
  }
  for (int i = threadIdx.x; i < ns; i += 256) {
    double m, p;
    lms_table_entry(i, mud, pwd, &m, &p);
    mt[i] = m; pt[i] = p;
    if (mtc) { if (i < n8) mtc[canon_mt_index(J, i)] = m; if (i < n4) ptc[canon_pt_index(J, i)] = p; }
  }
  double sum = 0.0;
  for (int b = 0; b < ns; b += 256) {
    __syncthreads();
    chunk[threadIdx.x] = (b + (int)threadIdx.x < ns) ? pt[b + threadIdx.x] : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = ns - b < 256 ? ns - b : 256;
      for (int i = 0; i < m; i++) sum += chunk[i];
    }
  }
  if (threadIdx.x == 0) it.sum_powtab[s] = sum;
}

void launch_tables(hipStream_t s, WorkItem *d_items, int count, double *d_tab) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_tables, dim3(count, 4), dim3(256), 0, s, d_items, d_tab);
}

// ------------------------------------------------------------------ stage 1: OLS
// (build-time knob, tools/build_variant.sh: the panel kernels at three workgroups per CU by launch bounds were measured as a loss)
#ifndef SACAMD_EXP_PANEL_MINB
#define SACAMD_EXP_PANEL_MINB 1
#endif
template <int NL, int NMAX> constexpr int ols_minb() { return (NL == 256 && NMAX <= 64) ? SACAMD_EXP_PANEL_MINB : 1; }
template <int NL, int NMAX>
__global__ __launch_bounds__(NL, (ols_minb<NL, NMAX>())) void k_ols(const WorkItem *items, const int *idx, PcmView v, double *pbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (idx[blockIdx.x] < 0) return;                 // padding entry of the XCD-interleaved launch list (host.hip, xcd_interleave)
  const WorkItem &it = items[idx[blockIdx.x]];
  const ChanParam p = it.p;
  const int *self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
  const int *other = v.pcm + it.frame * v.frame_stride + it.ch_other * v.ch_stride + it.start;
  ExecDev<NL> ex;
  double *out = (it.pin_kept ? v.keep : pbuf) + it.off_pin;   // off_pin: where this item's p_lpc lives
  if constexpr (NL == 64) ols_stage_reg<ExecDev<NL>, NMAX>(ex, p, self, other, it.n, out, smem, v.prof);
  else if constexpr (NL == 256 && NMAX > 64) ols_stage_panel2<ExecDev<NL>, NMAX>(ex, p, self, other, it.n, out, smem, v.prof);
  else ols_stage_panel<ExecDev<NL>, NMAX>(ex, p, self, other, it.n, out, smem, v.prof);
}

// Packed one-wave kernel for regressors up to 32 taps (pred_ols_pack.h): G = 64 / GL work-items per wave.  List position of
// group g of workgroup w: within every block of 8 G positions workgroup w % 8 takes the positions == w % 8 (mod 8), so the
// items of a workgroup share the residue that xcd_interleave (host.hip) gave their frames -- and with it the XCD whose L2
// holds those frames' samples.
template <int NMAX, int GL>
__global__ __launch_bounds__(64, 2) void k_ols_pack(const WorkItem *items, const int *idx, int count, PcmView v, double *pbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int G = 64 / GL;
  ExecDev<64> ex;
  ExecDev<64>::Reg<OlsPackSlot> sl;
  ExecDev<64>::Reg<int> kreg;
  const int w = (int)blockIdx.x, g = (int)threadIdx.x / GL;
  const int pos = (w >> 3) * (8 * G) + g * 8 + (w & 7);
  const int ii = pos < count ? idx[pos] : -1;
  const WorkItem &it = items[ii < 0 ? 0 : ii];
  sl.v.p = &it.p;
  sl.v.self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
  sl.v.other = v.pcm + it.frame * v.frame_stride + it.ch_other * v.ch_stride + it.start;
  sl.v.out = (it.pin_kept ? v.keep : pbuf) + it.off_pin;
  sl.v.n = ii < 0 ? 0 : it.n;
  kreg.v = ii < 0 ? 0 : it.p.k;
  int kk = 0;
#pragma unroll
  for (int q = 0; q < G; q++) { const int a = ex.lane_geti(kreg, q * GL); kk = a > kk ? a : kk; }
  if (kk <= 0) return;                                   // nothing but padding entries
  ols_stage_pack<ExecDev<64>, NMAX, GL>(ex, sl, kk, smem);
}

template <int NMAX, int GL>
static void launch_ols_pack_c(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, PcmView v, double *d_p) {
  static std::atomic<unsigned long long> done{0};
  constexpr int G = 64 / GL;
  const size_t bytes = ols_pack_lds_bytes<NMAX, GL>();
  if (ensure_dyn_lds((const void *)k_ols_pack<NMAX, GL>, bytes, done) != hipSuccess) return;
  const int blocks = ((count + 8 * G - 1) / (8 * G)) * 8;
  hipLaunchKernelGGL((k_ols_pack<NMAX, GL>), dim3(blocks), dim3(64), bytes, s, d_items, d_idx, count, v, d_p);
}

template <int NL, int NMAX>
static void launch_ols_c(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, PcmView v, double *d_p) {
  static std::atomic<unsigned long long> done{0};
  const size_t bytes = NL == 64 ? OlsLdsFast::bytes(NMAX) : ((NL == 256 && NMAX > 64) ? ols_panel2_lds_bytes(NMAX) : ols_panel_lds_bytes(NMAX, NL / 64));
  if (ensure_dyn_lds((const void *)k_ols<NL, NMAX>, bytes, done) != hipSuccess) return;   // the launch below would fail too; hipGetLastError reports it
  hipLaunchKernelGGL((k_ols<NL, NMAX>), dim3(count), dim3(NL), bytes, s, d_items, d_idx, v, d_p);
}

// 33..64 taps on one wave, matrix 2D-cyclic over the lanes (pred_ols_grid.h): NB = blocks of 8 rows / columns.  158 / 188 / 219 / 256
// registers for NB = 5 .. 8 (the backward solve keeps its operands sixteen to a register, DPP row broadcast): three waves per SIMD for
// the 40-tap instance, two for the others; search and final pass run the same build.
// Round 6: the 64-tap instance needs 260 registers; at two waves per SIMD (256) the four missing ones were two solved weights of the
// backward substitution, stored to and reloaded from scratch inside that dependent chain (31 scratch instructions).  The final pass
// (LAT: k = 1, one factorisation per sample, a few items per SIMD at most) therefore runs it at one wave per SIMD, where it has them.
template <int NB, bool LAT> constexpr int grid_waves() { return NB <= 5 ? 3 : (NB == 8 && LAT ? 1 : 2); }
template <int NB, bool LAT = false>
__global__ __launch_bounds__(64, (grid_waves<NB, LAT>())) void k_ols_grid(const WorkItem *items, const int *idx, PcmView v, double *pbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (idx[blockIdx.x] < 0) return;                 // padding entry of the XCD-interleaved launch list
  const WorkItem &it = items[idx[blockIdx.x]];
  const ChanParam p = it.p;
  const int *self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
  const int *other = v.pcm + it.frame * v.frame_stride + it.ch_other * v.ch_stride + it.start;
  ExecDev<64> ex;
  double *out = (it.pin_kept ? v.keep : pbuf) + it.off_pin;
  ols_stage_grid<ExecDev<64>, NB>(ex, p, self, other, it.n, out, smem, v.prof);
}
template <int NB, bool LAT = false>
static void launch_ols_grid_c(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, PcmView v, double *d_p) {
  const size_t bytes = OlsLdsGrid::bytes(8 * NB);       // <= 24 KB: below the default dynamic-LDS limit
  hipLaunchKernelGGL((k_ols_grid<NB, LAT>), dim3(count), dim3(64), bytes, s, d_items, d_idx, v, d_p);
}

constexpr int kOlsPanelThreads = 256;   // panel width 4 (8 waves measured slower: one workgroup per CU, and a barrier-parked wave sharing the SIMD of wave 0 doubles the time of its serial solve)
// latency_bound: the final pass (k = 1: one factorisation per sample, one work-item per frame x channel).  Its 33..64-tap items
// take the four-wave panel kernel, whose per-sample latency is lower (28 vs 34 us at 48 taps); the search (thousands of items
// per launch) takes the one-wave kernel, whose throughput is 1.1 - 1.7 x higher.  SACAMD_OLS_FINAL_PANEL=0 / 1 overrides.
void launch_ols(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, int ols_class, PcmView v, double *d_p, bool latency_bound) {
  if (count <= 0) return;
  static_assert(kNumOlsClasses == 8 && kOlsClassMax[6] == 64 && kOlsClassMax[7] == 96, "instances below follow kOlsClassMax");
  static const int force = [] { const char *e = std::getenv("SACAMD_OLS_FINAL_PANEL"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
  const bool panel = force >= 0 ? (force == 1 && latency_bound) : latency_bound;
  // regressors up to 32 taps: several work-items per wave (pred_ols_pack.h); SACAMD_OLS_PACK=0 selects the one-item-per-wave kernel (A/B)
  static const bool pack = [] { const char *e = std::getenv("SACAMD_OLS_PACK"); return !(e && e[0] == '0'); }();
  // 33..64 taps: the 2D-cyclic one-wave kernel (round 5) for search AND final pass; SACAMD_OLS_GRID=0 selects the round-4 kernels (A/B)
  static const bool grid = [] { const char *e = std::getenv("SACAMD_OLS_GRID"); return !(e && e[0] == '0'); }();
  // 25..32 taps: the grid kernel as well (497 against 310 M item-steps/s saturated, 13.0 against 18.5 us per sample at k = 1); 17..24
  // taps: the packed kernel in the search (778 against 737 M item-steps/s), the grid kernel in the final pass (9.2 against 11.8 us per
  // sample) -- profiles/r05/ols_grid_short.txt.  SACAMD_OLS_GRID_SHORT=0 keeps the packed kernels (A/B).
  static const bool grid_short = [] { const char *e = std::getenv("SACAMD_OLS_GRID_SHORT"); return !(e && e[0] == '0'); }();
  if (grid && grid_short && ols_class == 2) { launch_ols_grid_c<4>(s, d_items, d_idx, count, v, d_p); return; }
  if (grid && grid_short && ols_class == 1 && latency_bound) { launch_ols_grid_c<3>(s, d_items, d_idx, count, v, d_p); return; }
  if (grid && ols_class >= 3 && ols_class <= 6) {
    switch (ols_class) {
      case 3: launch_ols_grid_c<5>(s, d_items, d_idx, count, v, d_p); break;
      case 4: launch_ols_grid_c<6>(s, d_items, d_idx, count, v, d_p); break;
      case 5: launch_ols_grid_c<7>(s, d_items, d_idx, count, v, d_p); break;
      default: if (latency_bound) launch_ols_grid_c<8, true>(s, d_items, d_idx, count, v, d_p); else launch_ols_grid_c<8>(s, d_items, d_idx, count, v, d_p); break;
    }
    return;
  }
  switch (ols_class) {
    case 0: if (pack) launch_ols_pack_c<16, 16>(s, d_items, d_idx, count, v, d_p); else launch_ols_c<64, 16>(s, d_items, d_idx, count, v, d_p); break;
    case 1: if (pack) launch_ols_pack_c<24, 32>(s, d_items, d_idx, count, v, d_p); else launch_ols_c<64, 24>(s, d_items, d_idx, count, v, d_p); break;
    case 2: if (pack) launch_ols_pack_c<32, 32>(s, d_items, d_idx, count, v, d_p); else launch_ols_c<64, 32>(s, d_items, d_idx, count, v, d_p); break;
    case 3: if (panel) launch_ols_c<kOlsPanelThreads, 40>(s, d_items, d_idx, count, v, d_p); else launch_ols_c<64, 40>(s, d_items, d_idx, count, v, d_p); break;
    case 4: if (panel) launch_ols_c<kOlsPanelThreads, 48>(s, d_items, d_idx, count, v, d_p); else launch_ols_c<64, 48>(s, d_items, d_idx, count, v, d_p); break;
    case 5: if (panel) launch_ols_c<kOlsPanelThreads, 56>(s, d_items, d_idx, count, v, d_p); else launch_ols_c<64, 56>(s, d_items, d_idx, count, v, d_p); break;
    case 6: if (panel) launch_ols_c<kOlsPanelThreads, 64>(s, d_items, d_idx, count, v, d_p); else launch_ols_c<64, 64>(s, d_items, d_idx, count, v, d_p); break;
    default: launch_ols_c<256, 96>(s, d_items, d_idx, count, v, d_p); break;   // 65..96: panel factorisation, two rows per lane
  }
}

// ------------------------------------------------------------------ stage 2: cascade
// Register-capacity classes of the cascade: taps up to (2048,1024,512,256) / (4096,..) on 256 lanes,
// (8192,4096,2048,1024) on 512 lanes with the middle class's slot counts.  The two larger ones are
// capped at 256 VGPRs (a few chain temporaries go to scratch): two 4-wave workgroups resp. one
// 8-wave workgroup per CU.
// Tap-slot layouts (slots of NL taps per stage).  0: the small class; 1: twice that; 2: the profile maximum
// (8192, 4096, 2048, 1024) on 512 lanes, a whole CU per work-item.  3 and 4 hold the same 30 slots as class 1
// split differently: the search often makes ONE late stage long (stage 1 up to 3072, stage 2 up to 1280,
// stage 3 up to 768 taps) or stage 0 alone (up to 5120), which class 1's per-stage caps would send to class 2.
using LmsA = LmsClass<8, 4, 2, 1>;
using LmsB = LmsClass<16, 8, 4, 2>;       // (the 512-lane whole-CU layout, class 2: the profile's box maximum)
// The three 30-slot layouts of 256 lanes (classes 1, 3, 4).  Round 6: re-chosen for the items the five 15- / 22-slot layouts leave over (1 495 of the
// 23 045 items of a 128-frame search, gpurun_out/r05/search_vn_128.npy): the best triple by exhaustive search over the 60 widest-covering splits takes
// 1 463 of them, where (16,8,4,2), (10,12,5,3), (20,4,5,1) took 1 292 -- the rest falls to the whole-CU class 2, whose workgroups wait for drained CUs
// at the end of every generation (203 -> 32 items).
using LmsB2 = LmsClass<12, 12, 4, 2>;
using LmsD = LmsClass<18, 6, 4, 2>;
using LmsE = LmsClass<15, 8, 6, 1>;
// 5 and 6: 22 slots, the most that still fits 256 VGPRs without scratch overflow, split for a long stage 1
// or a long stage 0: together they take about 60 % of the search's cascade work (need histogram of the
// default bench run), leaving ~10 % to the 30-slot layouts
using LmsX = LmsClass<6, 10, 4, 2>;
using LmsY = LmsClass<13, 5, 3, 1>;
// 14 and 15 (round 6): two more 22-slot splits.  With the factored step-size table the 22-slot layouts run three workgroups per CU and
// ~6.9 k cycles per sample, the 30-slot layouts two and ~10.9 k; of the items no other 22-slot split holds (stage lengths of 23 045
// search items, gpurun_out/r05/search_vn_128.npy: 2 868 of them) these two take 48 %: a long stage 0 beside a long stage 2, and three
// mid-sized late stages.
using LmsF = LmsClass<14, 2, 4, 2>;
using LmsG = LmsClass<9, 5, 5, 3>;
// canonical-order layouts of the final pass (pred_lms.h, CANON): odd slot counts (bank-conflict-free strided
// ring reads), 256 lanes; 7: (2304, 1280, 768, 256) taps in one round over the lanes, 8: twice that in two rounds,
// 9: four times (covers the profile maximum)
using LmsK = LmsClass<9, 5, 3, 1>;
using LmsL = LmsClass<17, 9, 5, 3>;
// All canonical layouts read mutab / powtab from global memory (mode 2; L2-resident, one read per tap and sample): with
// the tables in LDS (mode 1, 27 instead of 9 bytes per tap) only one workgroup fits a CU from ~4000 taps on and launches
// had to be cut wherever the combined ring sizes of their items exceeded the LDS -- 56 launches serialised on four
// streams in the 384 x 20 s run.  Measured per-step time is the same in both modes.
constexpr int lms_canon_mode(int cls) { return (cls < kLmsCanonFirst || cls >= 14) ? 0 : (cls < kLmsCanon3First ? 2 : 3); }
// Lane-map canonical layouts (pred_lms.h, CANON 3): LmsClass<J,0,0,0> = J chain positions per lane; 256 lanes = 2 dot + 2
// power-sum waves, 512 lanes = 4 + 4.  An item takes the first of these its chains fit (canon3_fits); what fits none of them
// (more than ~7.5 k taps) falls back to the systolic four-round layout 9.
using LmsP17 = LmsClass<17, 0, 0, 0>;
using LmsP33 = LmsClass<33, 0, 0, 0>;
using LmsP49 = LmsClass<49, 0, 0, 0>;
template <int CLS> struct LmsCfg;
// Residency (workgroups per CU the compiler must make room for: __launch_bounds__) of the search cascade layouts.  Build-time knobs
// of tools/build_variant.sh (A/B libraries); the defaults are the measured optimum (profiles/r04/README.md 2): three per CU for the
// 15-slot layout (168 VGPRs, 42 spilled: 179 -> 230 M item-steps/s), two for the others (three per CU there spills 130-230 registers and
// loses 2-3x).
#ifndef SACAMD_EXP_LMS0_MINB
#define SACAMD_EXP_LMS0_MINB 3
#endif
template <> struct LmsCfg<0> { static constexpr int ROUNDS = 1; using C = LmsA; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS0_MINB; };
#ifndef SACAMD_EXP_LMS134_MINB
#define SACAMD_EXP_LMS134_MINB 2
#endif
#ifndef SACAMD_EXP_LMS56_MINB
#define SACAMD_EXP_LMS56_MINB 3        // round 6: factored step-size table (pred_lms.h) -> 166 registers, 64-sample staging -> 52.4 KB of LDS: three per CU
#endif
template <> struct LmsCfg<1> { static constexpr int ROUNDS = 1; using C = LmsB2; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS134_MINB; };
template <> struct LmsCfg<2> { static constexpr int ROUNDS = 1; using C = LmsB; static constexpr int NL = 512, MINB = 1; };
template <> struct LmsCfg<3> { static constexpr int ROUNDS = 1; using C = LmsD; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS134_MINB; };
template <> struct LmsCfg<4> { static constexpr int ROUNDS = 1; using C = LmsE; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS134_MINB; };
template <> struct LmsCfg<5> { static constexpr int ROUNDS = 1; using C = LmsX; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS56_MINB; };
template <> struct LmsCfg<6> { static constexpr int ROUNDS = 1; using C = LmsY; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS56_MINB; };
template <> struct LmsCfg<14> { static constexpr int ROUNDS = 1; using C = LmsF; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS56_MINB; };
template <> struct LmsCfg<15> { static constexpr int ROUNDS = 1; using C = LmsG; static constexpr int NL = 256, MINB = SACAMD_EXP_LMS56_MINB; };
template <> struct LmsCfg<7> { using C = LmsK; static constexpr int NL = 256, MINB = 2, ROUNDS = 1; };
template <> struct LmsCfg<8> { using C = LmsK; static constexpr int NL = 256, MINB = 2, ROUNDS = 2; };
template <> struct LmsCfg<9> { using C = LmsK; static constexpr int NL = 256, MINB = 1, ROUNDS = 4; };
template <> struct LmsCfg<10> { using C = LmsP17; static constexpr int NL = 256, MINB = 2, ROUNDS = 1; };   // (round 6: three per CU = 168 registers spilled 56 of them into the sample loop)
template <> struct LmsCfg<11> { using C = LmsP33; static constexpr int NL = 256, MINB = 2, ROUNDS = 1; };
template <> struct LmsCfg<12> { using C = LmsP49; static constexpr int NL = 256, MINB = 1, ROUNDS = 1; };
template <> struct LmsCfg<13> { using C = LmsP33; static constexpr int NL = 512, MINB = 2, ROUNDS = 1; };

template <int CLS>
__global__ __launch_bounds__(LmsCfg<CLS>::NL, LmsCfg<CLS>::MINB) void k_lms(const WorkItem *items, const int *idx, PcmView v, const double *tab, const double *pbuf, double *qbuf, LmsRingCap rc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NL = LmsCfg<CLS>::NL;
  using C = typename LmsCfg<CLS>::C;
  if (idx[blockIdx.x] < 0) return;                 // padding entry of the XCD-interleaved launch list
  const WorkItem &it = items[idx[blockIdx.x]];
  const ChanParam &p = it.p;   // read through the scalar cache (uniform address); a private copy would live in scratch
  double sp[4];
  for (int s = 0; s < 4; s++) sp[s] = it.sum_powtab[s];
  const int *self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
  ExecDev<NL> ex;
  lms_stage<ExecDev<NL>, C, lms_canon_mode(CLS), LmsCfg<CLS>::ROUNDS>(ex, p, sp, tab + it.off_tab, self, it.n, (it.pin_kept ? v.keep : pbuf) + it.off_pin, qbuf + it.off_p, smem, rc.c, v.prof, nullptr, it.off_tabc >= 0 ? tab + it.off_tabc : nullptr);
}

template <int CLS>
static void launch_lms_c(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, LmsRingCap rc, PcmView v, const double *d_tab, const double *d_p, double *d_q) {
  constexpr int NL = LmsCfg<CLS>::NL;
  using C = typename LmsCfg<CLS>::C;
  static std::atomic<unsigned long long> done{0};
  constexpr int CANON = lms_canon_mode(CLS);
  constexpr size_t kLdsPerCu = 160 * 1024;     // a layout's register capacity may exceed what one CU's LDS can hold as history
  const size_t full = CANON == 3 ? kLdsPerCu : LmsLds<NL, C, CANON>::bytes();
  if (ensure_dyn_lds((const void *)k_lms<CLS>, full < kLdsPerCu ? full : kLdsPerCu, done) != hipSuccess) return;
  const size_t bytes = LmsLds<NL, C, CANON>::bytes(rc.c);
  hipLaunchKernelGGL((k_lms<CLS>), dim3(count), dim3(NL), bytes, s, d_items, d_idx, v, d_tab, d_p, d_q, rc);
}

size_t lms_lds_bytes(int lms_class, const LmsRingCap &rc) {
  switch (lms_class) {
    case 0: return LmsLds<256, LmsA>::bytes(rc.c);
    case 1: return LmsLds<256, LmsB2>::bytes(rc.c);
    case 3: return LmsLds<256, LmsD>::bytes(rc.c);
    case 4: return LmsLds<256, LmsE>::bytes(rc.c);
    case 5: return LmsLds<256, LmsX>::bytes(rc.c);
    case 6: return LmsLds<256, LmsY>::bytes(rc.c);
    case 7: return LmsLds<256, LmsK, 2>::bytes(rc.c);
    case 8: return LmsLds<256, LmsK, 2>::bytes(rc.c);
    case 9: return LmsLds<256, LmsK, 2>::bytes(rc.c);
    case 10: return LmsLds<256, LmsP17, 3>::bytes(rc.c);
    case 11: return LmsLds<256, LmsP33, 3>::bytes(rc.c);
    case 12: return LmsLds<256, LmsP49, 3>::bytes(rc.c);
    case 13: return LmsLds<512, LmsP33, 3>::bytes(rc.c);
    case 14: return LmsLds<256, LmsF>::bytes(rc.c);
    case 15: return LmsLds<256, LmsG>::bytes(rc.c);
    default: return LmsLds<512, LmsB>::bytes(rc.c);
  }
}

// first layout, in order of cost, whose per-stage slots hold the item's stage lengths
int lms_class_for(const int *vn, bool canon) {
  auto fits = [&](int nl, int c0, int c1, int c2, int c3) { return vn[0] <= c0 * nl && vn[1] <= c1 * nl && vn[2] <= c2 * nl && vn[3] <= c3 * nl; };
  if (canon) {
    { const int c3 = canon3_class_for(vn); if (c3 >= 0) return c3; }      // lane-map layouts: chains fit the lanes, rings + tables one CU's LDS
    // beyond ~7.5 k taps: the round-2 systolic layouts
    if (fits(256, LmsK::c0, LmsK::c1, LmsK::c2, LmsK::c3)) return 7;     // one round over the lanes
    if (fits(512, LmsK::c0, LmsK::c1, LmsK::c2, LmsK::c3)) return 8;     // two rounds
    return 9;                                                            // four rounds (profile maximum)
  }
  if (fits(256, LmsA::c0, LmsA::c1, LmsA::c2, LmsA::c3)) return 0;
  if (fits(256, LmsX::c0, LmsX::c1, LmsX::c2, LmsX::c3)) return 5;
  if (fits(256, LmsY::c0, LmsY::c1, LmsY::c2, LmsY::c3)) return 6;
  if (fits(256, LmsF::c0, LmsF::c1, LmsF::c2, LmsF::c3)) return 14;
  if (fits(256, LmsG::c0, LmsG::c1, LmsG::c2, LmsG::c3)) return 15;
  if (fits(256, LmsB2::c0, LmsB2::c1, LmsB2::c2, LmsB2::c3)) return 1;
  if (fits(256, LmsD::c0, LmsD::c1, LmsD::c2, LmsD::c3)) return 3;
  if (fits(256, LmsE::c0, LmsE::c1, LmsE::c2, LmsE::c3)) return 4;
  return 2;
}

// register-file bound on resident workgroups per CU (237 / 256 / 256 registers, 4 / 4 / 8 waves)
int lms_max_wg_per_cu(int lms_class) {
  switch (lms_class) {
    case 0: return LmsCfg<0>::MINB > 2 ? LmsCfg<0>::MINB : 2;
    case 1: return LmsCfg<1>::MINB; case 3: return LmsCfg<3>::MINB; case 4: return LmsCfg<4>::MINB;
    case 5: return LmsCfg<5>::MINB; case 6: return LmsCfg<6>::MINB; case 14: return LmsCfg<14>::MINB; case 15: return LmsCfg<15>::MINB;
    case 10: return LmsCfg<10>::MINB;
    case 2: case 9: case 12: case 13: return 1;
    default: return 2;
  }
}

void launch_lms(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, int lms_class, LmsRingCap rc, PcmView v,
                const double *d_tab, const double *d_p, double *d_q) {
  if (count <= 0) return;
  switch (lms_class) {
    case 0: launch_lms_c<0>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 1: launch_lms_c<1>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 3: launch_lms_c<3>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 4: launch_lms_c<4>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 5: launch_lms_c<5>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 6: launch_lms_c<6>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 7: launch_lms_c<7>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 8: launch_lms_c<8>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 9: launch_lms_c<9>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 10: launch_lms_c<10>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 11: launch_lms_c<11>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 12: launch_lms_c<12>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 13: launch_lms_c<13>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 14: launch_lms_c<14>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    case 15: launch_lms_c<15>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
    default: launch_lms_c<2>(s, d_items, d_idx, count, rc, v, d_tab, d_p, d_q); break;
  }
}

// ------------------------------------------------------------------ stage 3: bias + residual
constexpr int kBiasSlabStride = kBiasSlabDoubles + 1;   // odd stride: spread lanes over LDS banks

__global__ __launch_bounds__(64) void k_bias(const WorkItem *items, int count, PcmView v, const FrameStatsD *stats, int nch,
                                              const double *pbuf, int *errbuf, int *predbuf, int *nonfinite, double *pdbuf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= count) return;
  const WorkItem &it = items[i];
  const ChanParam p = it.p;
  const int *self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
  double *tables = reinterpret_cast<double *>(smem) + (size_t)threadIdx.x * kBiasSlabStride;
  const int mean = stats[it.frame * nch + it.ch_self].mean;
  bias_stage(p, self, it.n, pbuf + it.off_p, mean, errbuf + it.off_err, predbuf ? predbuf + it.off_err : nullptr, tables, nonfinite ? nonfinite + i : nullptr, nullptr,
             pdbuf ? pdbuf + it.off_p : nullptr);
}

void launch_bias(hipStream_t s, const WorkItem *d_items, int count, PcmView v, const FrameStatsD *d_stats, int nch,
                 const double *d_p, int *d_err, int *d_pred, int *d_nonfinite, double *d_pd) {
  if (count <= 0) return;
  const size_t bytes = (size_t)64 * kBiasSlabStride * sizeof(double);
  hipLaunchKernelGGL(k_bias, dim3((count + 63) / 64), dim3(64), bytes, s, d_items, count, v, d_stats, nch, d_p, d_err, d_pred, d_nonfinite, d_pd);
}

// ------------------------------------------------------------------ decoder: the three stages of a channel side by side
// FrameCoder::UnpredictFrame (libsac.cpp:144-199): a sample is only known once it has been predicted, so the stages cannot
// run one after the other as in the encoder.  Each stage keeps its body (k = 1, canonical order) and all of them -- for both
// channels of a frame -- run AT THE SAME TIME, handing values over sample by sample through release / acquire counters in
// global memory (DecLink, simt.h).  Concurrent kernels need concurrent hardware queues, of which HIP maps only a few (and
// which streams share one is not ours to choose), so the whole group is TWO launches: k_dec_cascade (every cascade layout
// behind one entry point, one workgroup per channel) and k_dec_olsbias (OLS workgroups, then bias workgroups).  The host
// launches only as many frames at once as are certainly co-resident, and every wait is bounded: a missing partner ends
// in an error, not in a hang.
constexpr int kDecThreads = 256;
int dec_lms_class_for(const int *vn) {        // layouts of 256 lanes only (one block size for the whole launch)
  if (canon3_fits(vn, LmsP17::c0, 2)) return 10;
  if (canon3_fits(vn, LmsP33::c0, 2)) return 11;
  if (canon3_fits(vn, LmsP49::c0, 2)) return 12;
  return 9;
}
template <int CLS>
static __device__ __forceinline__ void dec_cascade_role(const WorkItem &it, const int *self, const double *tab, const double *pbuf, double *qbuf, char *smem,
                                                         const LmsRingCap &rc, const DecLink *link) {
  using C = typename LmsCfg<CLS>::C;
  double sp[4];
  for (int s = 0; s < 4; s++) sp[s] = it.sum_powtab[s];
  ExecDev<kDecThreads> ex;
  // every block lays out its history rings for ITS item's stage lengths (a decoder block is one work-item and owns its CU's
  // LDS): the launch asks for the largest single footprint, not for the per-stage maximum over all its items
  (void)rc;
  int own[4];
  for (int s = 0; s < 4; s++) own[s] = it.p.vn[s] + 1;
  lms_stage<ExecDev<kDecThreads>, C, lms_canon_mode(CLS), LmsCfg<CLS>::ROUNDS>(ex, it.p, sp, tab + it.off_tab, self, it.n, pbuf + it.off_pin, qbuf + it.off_p, smem, own, nullptr,
                                                                                link, it.off_tabc >= 0 ? tab + it.off_tabc : nullptr);
}
__global__ __launch_bounds__(kDecThreads, 1) void k_dec_cascade(const WorkItem *items, const int *idx, PcmView v, const double *tab, const double *pbuf, double *qbuf,
                                                                 LmsRingCap rc, const DecLink *links, int *started) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ii = idx[blockIdx.x];
  const WorkItem &it = items[ii];
  const int *self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
  if (threadIdx.x == 0) { __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }   // "resident" (host-visible counter)
  switch (it.lms_class) {
    case 10: dec_cascade_role<10>(it, self, tab, pbuf, qbuf, smem, rc, links + ii); break;
    case 11: dec_cascade_role<11>(it, self, tab, pbuf, qbuf, smem, rc, links + ii); break;
    case 12: dec_cascade_role<12>(it, self, tab, pbuf, qbuf, smem, rc, links + ii); break;
    default: dec_cascade_role<9>(it, self, tab, pbuf, qbuf, smem, rc, links + ii); break;
  }
}
size_t dec_cascade_lds_bytes(int lms_class, const LmsRingCap &rc) { return lms_lds_bytes(lms_class, rc); }
void launch_dec_cascade(hipStream_t s, const WorkItem *d_items, const int *d_idx, int count, size_t lds_bytes, LmsRingCap rc, PcmView v,
                        const double *d_tab, const double *d_p, double *d_q, const DecLink *d_links, int *d_started) {
  if (count <= 0) return;
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_lds((const void *)k_dec_cascade, 160 * 1024, done) != hipSuccess) return;
  hipLaunchKernelGGL(k_dec_cascade, dim3(count), dim3(kDecThreads), lds_bytes, s, d_items, d_idx, v, d_tab, d_p, d_q, rc, d_links, d_started);
}

// blocks [0, n_ols): the OLS stage of item idx[b] (one wave up to 64 taps -- the other three waves leave at once -- four waves
// beyond); blocks [n_ols, n_ols + n_bias): the bias stage of item idx[b] (one lane)
__global__ __launch_bounds__(kDecThreads) void k_dec_olsbias(const WorkItem *items, const int *idx, int n_ols, PcmView v, double *pbuf, const double *qbuf,
                                                              const FrameStatsD *stats, int nch, const DecLink *lk_ols, const DecLink *lk_bias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ii = idx[blockIdx.x];
  const WorkItem &it = items[ii];
  if ((int)blockIdx.x < n_ols) {
    const int *self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
    const int *other = v.pcm + it.frame * v.frame_stride + it.ch_other * v.ch_stride + it.start;
    if (it.p.n_ols <= 64) {
      if (threadIdx.x >= 64) return;
      ExecDev<64> ex;
      ols_stage_reg<ExecDev<64>, 64>(ex, it.p, self, other, it.n, pbuf + it.off_pin, smem, nullptr, lk_ols + ii);
    } else {
      ExecDev<256> ex;
      ols_stage_panel2<ExecDev<256>, 96>(ex, it.p, self, other, it.n, pbuf + it.off_pin, smem, nullptr, lk_ols + ii);
    }
  } else {
    if (threadIdx.x != 0) return;                         // one scalar recurrence: a wave of them would wait for each other's partners
    const int mean = stats[it.frame * nch + it.ch_self].mean;
    bias_stage(it.p, nullptr, it.n, qbuf + it.off_p, mean, nullptr, nullptr, reinterpret_cast<double *>(smem), nullptr, lk_bias + ii);
  }
}
void launch_dec_olsbias(hipStream_t s, const WorkItem *d_items, const int *d_idx, int n_ols, int n_bias, bool any_wide, PcmView v, double *d_p, const double *d_q,
                        const FrameStatsD *d_stats, int nch, const DecLink *d_lk_ols, const DecLink *d_lk_bias) {
  if (n_ols + n_bias <= 0) return;
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_lds((const void *)k_dec_olsbias, ols_panel2_lds_bytes(96), done) != hipSuccess) return;
  const size_t bytes = any_wide ? ols_panel2_lds_bytes(96) : OlsLdsFast::bytes(64);
  hipLaunchKernelGGL(k_dec_olsbias, dim3(n_ols + n_bias), dim3(kDecThreads), bytes, s, d_items, d_idx, n_ols, v, d_p, d_q, d_stats, nch, d_lk_ols, d_lk_bias);
}

// One-launch form of the same group: blocks [0, m) cascade, [m, 2m) OLS, [2m, 3m) bias of the items idx[b].  Every block is
// compiled for the cascade's register budget and asks for the largest LDS of the three roles, so each takes a CU of its own
// (3 m <= the CUs of the chip: the host sizes the group) -- and since all blocks of ONE grid that fits the chip are resident
// together, the stages can wait for each other without any assumption about concurrent hardware queues.
__global__ __launch_bounds__(kDecThreads, 1) void k_dec_all(const WorkItem *items, const int *idx, int m, PcmView v, const double *tab, double *pbuf, double *qbuf,
                                                             const FrameStatsD *stats, int nch, LmsRingCap rc, const DecLink *lk_lms, const DecLink *lk_ols,
                                                             const DecLink *lk_bias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = (int)blockIdx.x;
  const int ii = idx[b];
  const WorkItem &it = items[ii];
  const int *self = v.pcm + it.frame * v.frame_stride + it.ch_self * v.ch_stride + it.start;
  if (b < m) {
    switch (it.lms_class) {
      case 10: dec_cascade_role<10>(it, self, tab, pbuf, qbuf, smem, rc, lk_lms + ii); break;
      case 11: dec_cascade_role<11>(it, self, tab, pbuf, qbuf, smem, rc, lk_lms + ii); break;
      case 12: dec_cascade_role<12>(it, self, tab, pbuf, qbuf, smem, rc, lk_lms + ii); break;
      default: dec_cascade_role<9>(it, self, tab, pbuf, qbuf, smem, rc, lk_lms + ii); break;
    }
  } else if (b < 2 * m) {
    const int *other = v.pcm + it.frame * v.frame_stride + it.ch_other * v.ch_stride + it.start;
    if (it.p.n_ols <= 64) {
      if (threadIdx.x >= 64) return;
      ExecDev<64> ex;
      ols_stage_reg<ExecDev<64>, 64>(ex, it.p, self, other, it.n, pbuf + it.off_pin, smem, nullptr, lk_ols + ii);
    } else {
      ExecDev<256> ex;
      ols_stage_panel2<ExecDev<256>, 96>(ex, it.p, self, other, it.n, pbuf + it.off_pin, smem, nullptr, lk_ols + ii);
    }
  } else {
    if (threadIdx.x != 0) return;
    const int mean = stats[it.frame * nch + it.ch_self].mean;
    bias_stage(it.p, nullptr, it.n, qbuf + it.off_p, mean, nullptr, nullptr, reinterpret_cast<double *>(smem), nullptr, lk_bias + ii);
  }
}
void launch_dec_all(hipStream_t s, const WorkItem *d_items, const int *d_idx, int m, size_t lds_cascade, bool any_wide, LmsRingCap rc, PcmView v,
                    const double *d_tab, double *d_p, double *d_q, const FrameStatsD *d_stats, int nch, const DecLink *d_lk_lms, const DecLink *d_lk_ols,
                    const DecLink *d_lk_bias) {
  if (m <= 0) return;
  static std::atomic<unsigned long long> done{0};
  if (ensure_dyn_lds((const void *)k_dec_all, 160 * 1024, done) != hipSuccess) return;
  // at least 81 KB: no two blocks on one CU, whatever their roles need
  const size_t bytes = std::max({lds_cascade, any_wide ? ols_panel2_lds_bytes(96) : OlsLdsFast::bytes(64), (size_t)81 * 1024});
  // cooperative launch: the runtime admits the grid only if all 3 m blocks can be resident together -- the property the stage
  // blocks' waits rest on -- and fails the launch up front otherwise (round-3 advice), instead of leaving it to the bounded waits
  void *args[] = {(void *)&d_items, (void *)&d_idx, (void *)&m, (void *)&v, (void *)&d_tab, (void *)&d_p, (void *)&d_q, (void *)&d_stats, (void *)&nch,
                  (void *)&rc, (void *)&d_lk_lms, (void *)&d_lk_ols, (void *)&d_lk_bias};
  (void)hipLaunchCooperativeKernel((const void *)k_dec_all, dim3(3 * m), dim3(kDecThreads), args, (unsigned)bytes, s);
}

// prefix[j] = number of used values in [-32768, -32768 + j - 1], j = 0 .. 65537 (Remap::isUsed: 0 is always used), one block per job
__global__ __launch_bounds__(256) void k_used_prefix(const unsigned char *used, const long long *off_used, int *prefix) {
  __shared__ int part[256];
  const unsigned char *u = used + off_used[blockIdx.x];
  int *pf = prefix + (size_t)blockIdx.x * 65540;
  auto flag = [&](int v) { return v == 0 ? 1 : (v > 0 ? (int)u[32769 + v] : (int)u[-v]); };
  constexpr int N = 65537, PER = (N + 255) / 256;
  const int j0 = threadIdx.x * PER, j1 = (j0 + PER < N) ? j0 + PER : N;
  int loc = 0;
  for (int j = j0; j < j1; j++) loc += flag(j - 32768);
  part[threadIdx.x] = loc;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 256; i++) { const int t = part[i]; part[i] = run; run += t; } }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int j = j0; j < j1; j++) { pf[j] = run; run += flag(j - 32768); }
  if (j1 == N && j0 < N) pf[N] = run;
}
void launch_used_prefix(hipStream_t s, int count, const unsigned char *d_used, const long long *d_off_used, int *d_prefix) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_used_prefix, dim3(count), dim3(256), 0, s, d_used, d_off_used, d_prefix);
}

}  // namespace sacamd
