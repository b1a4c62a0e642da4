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
// Round 6: the search sweep (lms_sweep) is one branch-free software pipeline over all slots of the layout with the history loads a few
// slots ahead; the 22-slot layouts keep the step-size table factored (lms_mtfac); the RLS stage lives in wave 2's registers (rls_dispatch);
// the mixer's wave-uniform state and the loop-invariant parameters live in LDS -- DESIGN.md 4, "The search cascade since round 6".
//
// Arithmetic: the per-tap element operations are the reference's (explicit fma); the small dots of
// the serial chain use the canonical order (canon.h).  The N-term dot / power sums come in two forms:
//   CANON = false (search evaluations, k = optk): reduced in lane-then-tree order, NOT slmath::dot's
//     AVX2 order -- a ~1e-14 relative perturbation of p_lms that only moves search costs.
//   CANON = true (the final pass, whose residual goes into the bitstream and which the reference
//     decoder recomputes): exactly slmath::dot / calc_s2pow (common/math.h:130-191): 8 resp. 4
//     strided serial FMA chains, (s1+s2), ((b0+b1)+b2)+b3, then the transform_reduce tail -> p_lms is
//     bit-identical to the reference's.  Layout for that: the taps of chain c are spread over the
//     lanes of one wave in blocks of J consecutive chain positions (lane m holds positions
//     mJ..mJ+J-1), the running sum hops from lane to lane (DPP wave_shr:1) and every lane adds its J
//     terms with dependent FMAs -- a systolic pass whose depth is the chain length, which is the
//     floor any implementation of that summation order has.  The history rings are padded by one
//     element per eight so that the strided reads are bank-conflict free for odd J.
#pragma once
#include <type_traits>
#include "canon.h"
#include "libm_port.h"
#include "params.h"
#include "pred_tables.h"

namespace sacamd {

template <int C0, int C1, int C2, int C3>
struct LmsClass {
  static constexpr int c0 = C0, c1 = C1, c2 = C2, c3 = C3;
  static constexpr int total = C0 + C1 + C2 + C3;
  SA_HD static constexpr int slots(int s) { return s == 0 ? C0 : s == 1 ? C1 : s == 2 ? C2 : C3; }
  SA_HD static constexpr int first(int s) { return s == 0 ? 0 : s == 1 ? C0 : s == 2 ? C0 + C1 : C0 + C1 + C2; }
};
constexpr int kRlsMax = 10;
#ifndef SACAMD_EXP_LMS_AHEAD
#define SACAMD_EXP_LMS_AHEAD 3
#endif
// search sweep: slots whose history loads are in flight ahead of the arithmetic (4 registers per slot); the 30-slot layouts
// already fill the 256-VGPR budget with their tap state
template <class C> constexpr int lms_ahead() { return C::total >= 22 ? 2 : SACAMD_EXP_LMS_AHEAD; }

// Factored step-size table (round 6).  mutab[i] = mu_decay^i (ls.h:40) is one of the three doubles a tap keeps in registers.  With tap
// i = NL j + l it factors into a per-lane value ml[l] = mutab[l] (one register per STAGE) and a per-slot value Mj = mutab[NL j] (uniform:
// lane q of every wave keeps the factor of flat slot q in ONE register, the sweep takes it by v_readlane -- a per-slot LDS read was tried
// first and cost 17 cycles per slot in LDS pipe contention): the update becomes  w = fma(Mj, (wg ml) xo, w)  instead of  fma(mutab[i], wg xo, w) -- the same
// number of operations, two registers per tap less (22-slot layouts: 211 -> 166 registers = three workgroups per CU).  Mj ml differs from
// the table's correctly rounded mu_decay^i by ~1 ulp (exactly equal when mu_decay = 1, the default): a perturbation of the update far below
// the weight's own rounding, the same class as the free summation order of these search evaluations -- the final pass keeps the
// table.  Taps beyond the stage length need no zero entry any more: the history behind the window is KEPT at zero (the gains lane clears
// the value that leaves the window at every push), so their update term is 0 and their weight stays +0.  SACAMD_EXP_LMS_MTFAC=0: table.
#ifndef SACAMD_EXP_LMS_MTFAC
#define SACAMD_EXP_LMS_MTFAC 1
#endif
// Only where it buys residency: the 22-slot layouts (three workgroups per CU instead of two: +18 % saturated throughput).  The 15- and
// 30-slot layouts keep the table in registers -- factored they ran 5-6 % slower (two v_readlane per slot) at unchanged residency.
template <class C> SA_HD constexpr bool lms_mtfac() { return SACAMD_EXP_LMS_MTFAC != 0 && C::total == 22; }
template <int N> struct DArr { double v[N]; };

// f(std::integral_constant<int, I>{}) for I = 0 .. N-1: a loop whose index is a compile-time constant in every
// iteration (register arrays indexed with it are never addressed dynamically, whatever the unroller decides)
template <int I, int N, class F>
SA_HD void static_for(F &&f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// ---- CANON 3: lane-map layout of the canonical-order sums (round 3) ----
// All 48 summation chains of a sample (4 stages x (8 dot + 4 power-sum chains)) run AT THE SAME TIME, each on a run of
// adjacent lanes of one wave: a chain of K positions owns g = ceil(K / J) lanes, lane m of the run holds positions
// mJ .. mJ+J-1 (weights resp. powtab in registers for the whole frame, history values loaded once per sample) and the
// running sum hops from lane to lane (DPP wave_shr:1) exactly as in the older layouts.  Because lanes are handed out
// in proportion to chain length, every wave finishes after about max(K) dependent FMAs, and one FMA instruction
// carries 64 / g chains instead of the two (of one stage) the older layouts gave it.  Waves are homogeneous: the
// first D waves of the workgroup hold dot chains (they also update the weights), the other D hold power-sum chains.
// The lane -> (stage, chain, block) map depends on the item's stage lengths and is computed by every lane at start-up
// (canon3_pack: chains sorted by lane count, first fit into the waves of their kind; a chain never straddles a wave).
struct Canon3Lane { int st, ch, m, g; bool act; };
// kind 0 = dot chains (stride 8; at least one lane per chain: its first lane also owns the chain's tail tap),
// kind 1 = power-sum chains (stride 4).  q = lane index within the kind (0 .. 64*D-1) or -1 (fit test only).
// Returns false when the chains do not fit D waves of 64 lanes with J positions per lane.
SA_HD bool canon3_pack(const int *vn, int J, int D, int kind, int q, Canon3Lane *out, int *gmax) {
  int g[4], order[4] = {0, 1, 2, 3};
  for (int s = 0; s < 4; s++) {
    const int K = vn[s] >= 8 ? (kind ? vn[s] >> 2 : vn[s] >> 3) : 0;
    g[s] = (K + J - 1) / J;
    if (!kind && g[s] < 1) g[s] = 1;
    if (g[s] > 64) return false;
  }
  for (int a = 1; a < 4; a++)              // insertion sort, stable, longest first
    for (int b = a; b > 0 && g[order[b]] > g[order[b - 1]]; b--) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (out) { out->st = 0; out->ch = 0; out->m = 0; out->g = 0; out->act = false; }
  int gm = 0;
  const int nchain = kind ? 4 : 8;
  for (int a = 0; a < 4; a++) {
    const int s = order[a];
    if (!g[s]) continue;
    gm = g[s] > gm ? g[s] : gm;
    for (int c = 0; c < nchain; c++) {
      int w = 0;
      while (w < D && fill[w] + g[s] > 64) w++;
      if (w == D) return false;
      const int first = 64 * w + fill[w];
      if (out && q >= first && q < first + g[s]) { out->st = s; out->ch = c; out->m = q - first; out->g = g[s]; out->act = true; }
      fill[w] += g[s];
    }
  }
  if (gmax) *gmax = gm;
  return true;
}
SA_HD bool canon3_fits(const int *vn, int J, int D) {
  return canon3_pack(vn, J, D, 0, -1, nullptr, nullptr) && canon3_pack(vn, J, D, 1, -1, nullptr, nullptr);
}
// history rings of the lane-map layout: capacity >= 8J+8 and the first 8J+8 elements mirrored behind the end, so that a
// lane's window of J positions (stride 8 or 4) never wraps: one base address per lane and sample, immediate offsets
SA_HD constexpr int canon3_ext(int J) { return 8 * J + 8; }

// f(std::integral_constant<int, M>{}) for the run-time RLS order m = M in 1..kRlsMax (uniform branch; everything inside has compile-time sizes)
template <int M = 1, class F>
SA_HD void rls_dispatch(int m, F &&f) {
  if constexpr (M <= kRlsMax) { if (m == M) f(std::integral_constant<int, M>{}); else rls_dispatch<M + 1>(m, f); }
}
// P-row dot of the RLS stage with a run-time order 1..10 but compile-time unrolling
template <class A, class B>
SA_HD double dot_canon_m(int m, A a, B b) {
  switch (m) {
    case 1: return dot_canon_n<1>(a, b); case 2: return dot_canon_n<2>(a, b); case 3: return dot_canon_n<3>(a, b);
    case 4: return dot_canon_n<4>(a, b); case 5: return dot_canon_n<5>(a, b); case 6: return dot_canon_n<6>(a, b);
    case 7: return dot_canon_n<7>(a, b); case 8: return dot_canon_n<8>(a, b); case 9: return dot_canon_n<9>(a, b);
    default: return dot_canon_n<10>(a, b);
  }
}

// CANON: 0 = free summation order (search), 1 = slmath::dot order with mutab / powtab in LDS, 2 = the same with the
// tables read from global memory (profiles whose tables do not fit the CU's LDS beside the histories)
template <int NL, class C, int CANON = 0>
struct LmsLds {
  SA_HD static constexpr int ridx(int a) { return CANON ? a + (a >> 3) : a; }   // physical ring index
  // CANON 3: logical ring length for a capacity request c (>= taps + 1): capacity raised to the mirror length, plus the mirror
  SA_HD static constexpr int ringcap3(int c) { return c > canon3_ext(C::c0) ? c : canon3_ext(C::c0); }
  SA_HD static constexpr int ringlen(int c) { return CANON == 3 ? ringcap3(c) + canon3_ext(C::c0) : c; }
  double *csum, *psum, *tailw, *tailpw;   // CANON: chain sums [2][4][8], [2][4][4]; tail weights [2][4][8]; tail powtab [4][8]
  double *mt[4], *pt[4];                  // CANON: mutab / powtab of each stage, indexed like the rings (ridx(tap)); read once per sample
  double *ring[4];
  double *part;     // [2][NL/64][8]
  double *bc;       // [4]: wgrad of each stage
  double *pin, *pout;
  double *rph;               // round 6: the wave-uniform mixer state `us` (the RLS vectors live in wave 2's registers)
  double *P;                 // RLS inverse covariance, row l at P + l * kRlsMax (owned by the lanes of wave 2)
  double *pv;                // stage predictions p[0..3]
  double *exwm;              // expert weights mirror [2][5]
  double *cst;               // vmu[4], sum_powtab[4]; [8..15] proj_alpha, 1 - proj_alpha, mu_mix, mu_mix_beta, 1 - mu_mix_beta, lm_alpha, lo, hi (round 6: the
                             // work-item's ChanParam may alias the kernel's output for all the compiler knows, so p.x inside the sample loop was a
                             // global_load per sample in the middle of the serial chains)
  double *hs;                // head -> pieces: target, ep[2], pl[5], bp4, rpx; [10] next RLS prediction (wave 2 -> head); [12..13] blend weights smw
  double *libm;              // staged log/exp tables of libm_port.h
  int *sv;
  // ringcap[s] >= vn[s] + 1 of every work-item of the launch (<= C::slots(s) * NL + 1): the LDS
  // footprint follows the taps actually in use, not the register-capacity class
  // samples staged per global <-> LDS exchange (p_lpc in, p_lpc + p_lms out, the channel's samples): one per lane, except in the 22-slot
  // search layouts, where 64 keep the workgroup under a third of a CU's LDS (three workgroups per CU)
  SA_HD static constexpr int chunk() { return (CANON == 0 && C::total == 22 && NL == 256) ? 64 : NL; }
  // search layouts (CANON 0, round 6): the sweep visits every slot of the layout, so every ring has the layout's capacity
  SA_HD static constexpr int cap_of(int s, int c) { return CANON ? c : C::slots(s) * NL + 1; }
  SA_HD static size_t bytes(const int *ringcap) {
    size_t d = 0;
    #pragma unroll
    for (int s = 0; s < 4; s++) d += ((size_t)ridx(ringlen(cap_of(s, ringcap[s]))) + 1) * (CANON == 1 ? 3 : 1);      // + the mirror element ring[cap] == ring[0]; CANON 1: + mutab, powtab
    if (CANON) d += 64 + 32 + 64 + 32;
    if (CANON == 3) d += (size_t)(NL / 2) * C::c0;     // mutab of the dot lanes, lane-major per wave
    d -= 2 * (NL - chunk());                           // pin / pout hold one chunk
    d += 2 * (NL / 64) * 8 + 4 + 2 * NL + kRlsMax + kRlsMax * kRlsMax + 4 + 10 + 16 + 16 + kLibmLdsDoubles;   // part, bc[4], pin/pout, mixer state, RLS P, pv[4], exwm, cst, hs, libm   // pin/pout: NL samples staged per exchange
    return d * sizeof(double) + chunk() * sizeof(int) + 16;
  }
  SA_HD static size_t bytes() {
    const int full[4] = {C::slots(0) * NL + 1, C::slots(1) * NL + 1, C::slots(2) * NL + 1, C::slots(3) * NL + 1};
    return bytes(full);
  }
  SA_HD void carve(char *base, const int *ringcap) {
    double *d = reinterpret_cast<double *>(base);
    csum = psum = tailw = tailpw = nullptr;
    if (CANON) {
      // the chain-sum arrays come first: ring[0] must not sit at LDS offset 0 (an address formed as "element
      // i-1, immediate offset +8" would fall below the LDS aperture for i == 0 where the compiler uses a FLAT access)
      csum = d; d += 64; psum = d; d += 32; tailw = d; d += 64; tailpw = d; d += 32;
    }
    #pragma unroll
    for (int s = 0; s < 4; s++) { ring[s] = d; d += (size_t)ridx(ringlen(cap_of(s, ringcap[s]))) + 1; }
    #pragma unroll
    for (int s = 0; s < 4; s++) { mt[s] = pt[s] = nullptr; if (CANON == 1) { mt[s] = d; d += (size_t)ridx(ringcap[s]) + 1; pt[s] = d; d += (size_t)ridx(ringcap[s]) + 1; } }
    if (CANON == 3) { mt[0] = d; d += (size_t)(NL / 2) * C::c0; }
    part = d; d += 2 * (NL / 64) * 8;
    bc = d; d += 4;
    pin = d; d += chunk(); pout = d; d += chunk();
    rph = d; d += kRlsMax;
    P = d; d += kRlsMax * kRlsMax;
    pv = d; d += 4; exwm = d; d += 10; cst = d; d += 16; hs = d; d += 16;
    libm = d; d += kLibmLdsDoubles;
    sv = reinterpret_cast<int *>(d);
  }
};


// Lane-map layout (CANON 3) of an item with stage lengths vn: 10 / 11 / 12 = 17 / 33 / 49 positions per lane on 256 lanes, 13 = 33 on 512
// lanes, -1 = none.  The chains must fit the lanes (canon3_fits) AND the item's rings, mirrors and lane-major mutab block one CU's
// LDS: the 512-lane layout spends 68 KB on mutab alone and stops fitting near 7.5 k taps although its chains would hold 8.4 k (round 5:
// the reference's sequential search picks such profiles; the launch used to fail with "invalid argument").
SA_HD int canon3_class_for(const int *vn) {
  const int rc[4] = {vn[0] + 1, vn[1] + 1, vn[2] + 1, vn[3] + 1};
  constexpr size_t kLds = 160 * 1024;
  if (canon3_fits(vn, 17, 2) && LmsLds<256, LmsClass<17, 0, 0, 0>, 3>::bytes(rc) <= kLds) return 10;
  if (canon3_fits(vn, 33, 2) && LmsLds<256, LmsClass<33, 0, 0, 0>, 3>::bytes(rc) <= kLds) return 11;
  if (canon3_fits(vn, 49, 2) && LmsLds<256, LmsClass<49, 0, 0, 0>, 3>::bytes(rc) <= kLds) return 12;
  if (canon3_fits(vn, 33, 4) && LmsLds<512, LmsClass<33, 0, 0, 0>, 3>::bytes(rc) <= kLds) return 13;
  return -1;
}

// tail of slmath::dot handled by std::transform_reduce (< 8 elements), operands through getters
template <class A, class B>
SA_HD double tr_dot_g(int n, A a, B b) {
  double init = 0.0;
  int o = 0;
  if (n >= 4) {
    const double v1 = fma(a(1), b(1), a(0) * b(0));
    const double v2 = fma(a(3), b(3), a(2) * b(2));
    init = init + (v1 + v2);
    o = 4;
  }
  if (n - o >= 2) { init = init + a(o) * b(o); init = init + a(o + 1) * b(o + 1); o += 2; }
  if (o < n) init = fma(a(o), b(o), init);
  return init;
}
// same for calc_s2pow: terms pw * (x * x)
template <class A, class B>
SA_HD double tr_s2pow_g(int n, A x, B pw) {
  double init = 0.0;
  int o = 0;
  if (n >= 4) {
    const double x0 = x(0), x1 = x(1), x2 = x(2), x3 = x(3);
    const double v1 = fma(x1 * x1, pw(1), (x0 * x0) * pw(0));
    const double v2 = fma(x3 * x3, pw(3), (x2 * x2) * pw(2));
    init = init + (v1 + v2);
    o = 4;
  }
  if (n - o >= 2) { const double xa = x(o), xb = x(o + 1); init = init + (xa * xa) * pw(o); init = init + (xb * xb) * pw(o + 1); o += 2; }
  if (o < n) { const double xa = x(o); init = fma(xa * xa, pw(o), init); }
  return init;
}

// ---- search sweep (CANON 0), one stage of one lane: all SL tap slots of the layout, straight-line ----
// Round 6.  The sweep used to be one uniform branch per slot (slots beyond the stage length were skipped) with the slot's ring load,
// its `s_waitcnt lgkmcnt(0)` and its seven dependent fp64 operations inside: every slot paid a full LDS round trip (~300 cycles per
// 256-tap slot against ~40 of issue), and the branch conditions of 22 slots overflowed the scalar registers.  Now a stage's sweep has
// no branch at all: every slot of the layout is swept -- a tap beyond the stage length is arithmetically neutral, its mutab / powtab
// entries are zero and its weight stays +0: fma(0, ., w) = w, fma(x, 0, d) = d, fma(0, ., sp) = sp -- and the history loads of slot
// j + AHEAD are in flight under slot j's arithmetic.  For that every tap of the layout must read a finite value, so the rings of
// the search layouts have the LAYOUT's capacity (cap = SL NL + 1), not the item's (they are zero-filled at the start; LDS per
// workgroup is the layout's maximum, which the big launches of a batch search reached anyway).
// Addresses: slot j of lane l reads ring[(ps + l + NL j) mod cap] and its successor (ring[cap] mirrors ring[0]).  Two bases per lane
// and stage -- a0 = ring + ps + l and a1 = a0 - cap -- and one compare per slot (NL j >= cap - ps - l) select the wrapped one; the
// slot's offset NL j is a constant.  Operation order per tap and summation order over the slots are those of the rounds 1-5 sweep,
// and the neutral slots add exact zeros: results are bit-identical to it.
// The four stages' slots form ONE pipeline (flat slot q = C::first(s) + j): the loads of the next stage's first slots are in flight
// under the last slots of the stage before, so a sample exposes one LDS round trip, not four.
template <class C> SA_HD constexpr int lms_flat_stage(int q) { return q < C::first(1) ? 0 : (q < C::first(2) ? 1 : (q < C::first(3) ? 2 : 3)); }
template <int NL, class C, int AHEAD, class E, class T, class TM, class A8, class RM>
SA_HD void lms_sweep(E &ex, T &Wl, const TM &MTl, const T &PTl, A8 &accl, const double *ring0, const int *rofs, const int *pos, const double *bc, const RM &mjv, int l) {
  constexpr int TOT = C::total, G = AHEAD < TOT ? AHEAD : TOT;
  constexpr bool kLmsMtFac = lms_mtfac<C>();
  double bn[G], bo[G];
  const double *a0[4], *a1[4];
  int thr[4];
  double wg[4];
  auto load = [&](auto QC) {
    constexpr int q = decltype(QC)::value, s = lms_flat_stage<C>(q), j = q - C::first(s);
    constexpr int cp = C::slots(s) * NL + 1;
    if constexpr (j == 0) {                               // first slot of a stage: its bases and its gain
      const int u = pos[s] + l;
      a0[s] = ring0 + rofs[s] + u; a1[s] = a0[s] - cp; thr[s] = cp - u;      // slot j wraps iff NL j >= thr
      wg[s] = bc[s];
      if constexpr (kLmsMtFac) wg[s] = wg[s] * MTl.v[s];     // (wg ml): the lane's factor of the step sizes, once per stage
    }
    const double *a = (j * NL >= thr[s]) ? a1[s] : a0[s];
    bn[q % G] = a[j * NL]; bo[q % G] = a[j * NL + 1];
  };
  static_for<0, G>(load);
  double d = 0.0, sp = 0.0;
  static_for<0, TOT>([&](auto QC) {
    constexpr int q = decltype(QC)::value, s = lms_flat_stage<C>(q), j = q - C::first(s);
    if constexpr (j == 0) { d = 0.0; sp = 0.0; }
    const double xn = bn[q % G], xo = bo[q % G];
    double mq;
    if constexpr (kLmsMtFac) mq = ex.wave_lane(mjv, l, q); else mq = MTl.v[q];     // the slot's factor: lane q of every wave holds it (v_readlane -> a scalar operand of the fma)
    if constexpr (q + G < TOT) load(std::integral_constant<int, q + G>{});
    double w = fma(mq, wg[s] * xo, Wl.v[q]);
    w = clampd(w, -10.0, 10.0);
    Wl.v[q] = w;
    d = fma(xn, w, d);
    sp = fma(PTl.v[q], xn * xn, sp);
    SA_PIN_F64(d); SA_PIN_F64(sp);     // the slot's arithmetic stays in front of the fence ...
    SA_SCHED_FENCE();                  // ... across which the scheduler moves nothing (else it hoists every load of the sweep to its top: 4 registers per slot, spills)
    if constexpr (j == C::slots(s) - 1) { accl.v[s] = d; accl.v[4 + s] = sp; }
  });
}

template <class E, class C, int CANON = 0, int ROUNDS = 1>
SA_HD void lms_stage(E &ex, const ChanParam &p, const double *sum_powtab, const double *tab,
                     const int *self, int n, const double *pin_g, double *pout_g, char *lds_base, const int *ringcap,
                     unsigned long long *prof = nullptr, const DecLink *dec = nullptr, const double *tabc = nullptr) {
  constexpr int NL = E::nl;
  constexpr int NW = NL / 64;
  constexpr int kLmsChunk = LmsLds<NL, C, CANON>::chunk();   // samples staged per global<->LDS exchange
  constexpr bool kLmsMtFac = lms_mtfac<C>();      // search layouts: factored step-size table (see above)
  LmsLds<NL, C, CANON> L;
  L.carve(lds_base, ringcap);
  auto ridx = [](int a) { return LmsLds<NL, C, CANON>::ridx(a); };
  // CANON geometry: the 8 dot chains are spread CPW per wave over LPC lanes each; the 4 power-sum chains
  // run on waves 0..3, 64 lanes each, with SMUL times the slots per lane
  static_assert(!CANON || NL == 256 || NL == 512, "canonical layout: 4 or 8 waves");
  constexpr bool LM = CANON == 3;                 // lane-map layout (canon3_pack)
  constexpr int J3 = LM ? C::c0 : 1, D3 = NL / 128, EXT3 = canon3_ext(J3);
  // ROUNDS: a chain passes ROUNDS times over its lanes (positions r*LPC*J + m*J + j); the weights of all rounds stay in
  // registers, the chain operands of one round at a time.  The power-sum chains (twice the positions, twice the lanes
  // when NL = 256) make ROUNDS * SMUL rounds.
  constexpr int CPW = CANON ? 8 / NW : 1, LPC = 64 / CPW, SMUL = CANON ? NL / 256 : 1, RD = CANON ? ROUNDS : 1, RP = RD * SMUL;
  constexpr int NX = CANON ? C::slots(0) : 1;      // chain operands of ONE stage at a time; stage 0 has the most slots
  static_assert(!CANON || (C::c0 >= C::c1 && C::c0 >= C::c2 && C::c0 >= C::c3), "stage 0 holds the most slots");

  typename E::template Reg<DArr<C::total * RD>> W;                // CANON: slot (stage s, round r, j) at C::first(s) * RD + r * J + j
  typename E::template Reg<DArr<CANON ? 1 : (kLmsMtFac ? 4 : C::total)>> MT;   // search: mutab per tap, or (factored) ml of the lane per stage
  typename E::template Reg<double> mjv;                          // search, factored table: lane q of every wave holds Mj of flat slot q
  typename E::template Reg<DArr<CANON ? 1 : C::total>> PT;       // CANON: the tables stay in LDS (read once per sample): the chain operands need the registers
  typename E::template Reg<DArr<NX>> PR;                 // CANON: powtab of this lane's power-chain taps, loaded for the duration of the chains
  typename E::template Reg<DArr<8>> acc;
  typename E::template Reg<DArr<(CANON && !LM) ? 4 : 1>> Wt;        // CANON: the chain's tail tap (taps beyond 8*floor(n/8)) of lanes m == 0
  // lane-map layout: W = weights (dot lanes) resp. powtab (power lanes), XD = this sample's history values resp. squares
  typename E::template Reg<int> q_pos, q_cap, q_tap0, q_ro, q_sc, q_fl, q_ttap;   // ring position / capacity / first tap / ring offset / stage*8+chain / flags / tail tap
  typename E::template Reg<double> q_acc, q_mut;                                   // running chain sum; mutab of the tail tap
  int G3d = 0, G3p = 0;                                                           // hops of the dot / power waves
  typename E::template Reg<DArr<NX>> XD;                        // CANON: history values of this lane's dot-chain taps
  typename E::template Reg<DArr<NX>> XX;                 // CANON: squared history values of this lane's power-chain taps
  typename E::template Reg<double> sd[4], sq[4], hop;           // CANON: running chain sums

  int ns[4], cap[4], pos[4];
  #pragma unroll
  for (int s = 0; s < 4; s++) { ns[s] = p.vn[s]; cap[s] = CANON ? ns[s] + 1 : C::slots(s) * NL + 1; pos[s] = 0; if (LM && cap[s] < EXT3) cap[s] = EXT3; }
  if constexpr (LM) { canon3_pack(ns, J3, D3, 0, -1, nullptr, &G3d); canon3_pack(ns, J3, D3, 1, -1, nullptr, &G3p); }
  const int m = p.lm_n;

  // ---- init: tables -> registers, zero rings / weights
  // wave-0 lane roles of the mixer chain: lanes 0..9 = expert e=l/5, input i=l%5 (LS_ADA weight +
  // squared-gradient EMA in registers); lanes 0..m-1 = row l of the RLS inverse covariance P.
  // Per-lane state of the mixer chain.  Every wave uses its own few of these on its own lanes only, so
  // they share four registers (the tap state already fills the register file in the large classes);
  // the RLS matrix P lives in LDS and is held in registers only while wave 2 works on it.
  typename E::template Reg<double> mr0, mr1, mr2, mr3;
  auto &dots_r = mr0; auto &spow_r = mr1; auto &vmu_r = mr2; auto &spt_r = mr3;   // wave 0, lanes 16..19 (vmu / sum_powtab of the lane's stage: constants)
  auto &exw_r = mr0; auto &exeg_r = mr1;                                    // wave 1, lanes 0..9
  auto &rw_r = mr0; auto &ph_r = mr1; auto &x_r = mr2; auto &rcp_r = mr3;   // wave 2, lane l < lm_n: RLS weight w[l], (P x)[l], history x[l]; lanes 0, 1: reciprocals
  auto &exz_r = mr0;                                                        // wave 3
  ex.par([&](int l) {
    const double *tp = tab;
    mjv[l] = 0.0;
    #pragma unroll
    for (int s = 0; s < 4; s++) {
      const int f = C::first(s);
      if constexpr (!CANON) {
        for (int j = 0; j < C::slots(s); j++) {
          const int tap = j * NL + l;
          const bool on = tap < ns[s];
          W[l].v[f + j] = 0.0;
          if constexpr (!kLmsMtFac) MT[l].v[f + j] = on ? tp[tap] : 0.0;
          PT[l].v[f + j] = on ? tp[ns[s] + tap] : 0.0;
        }
        if constexpr (kLmsMtFac) {          // ml = mutab[l], Mj = mutab[NL j] (exact table entries; beyond the stage length: unused, 0)
          MT[l].v[s] = l < ns[s] ? tp[l] : 0.0;
          const int jq = (l & 63) - f;       // lane q = f + j of every wave: the factor of flat slot q
          if (jq >= 0 && jq < C::slots(s)) mjv[l] = jq * NL < ns[s] ? tp[jq * NL] : 0.0;
        }
      } else if constexpr (LM) {
        if (l < 8) { const int K4 = ns[s] >= 8 ? ns[s] >> 2 : 0; const int ti = 4 * K4 + l; L.tailpw[s * 8 + l] = ti < ns[s] ? tp[ns[s] + ti] : 0.0; }
      } else {
        // dot layout: lane (wave w, half, m) owns positions m*J..m*J+J-1 of chain c = w*CPW + half, i.e. taps 8k + c;
        // positions >= K8 = n/8 are empty.  The chain's tail tap 8*K8 + c (if < n) sits in the extra slot of lane m == 0.
        const int K4 = ns[s] >= 8 ? ns[s] >> 2 : 0;
        for (int j = 0; j < C::slots(s) * RD; j++) W[l].v[f * RD + j] = 0.0;
        Wt[l].v[s] = 0.0;
        if constexpr (CANON == 1) for (int i = l; i < ns[s]; i += NL) { L.mt[s][ridx(i)] = tp[i]; L.pt[s][ridx(i)] = tp[ns[s] + i]; }
        if (l < 8) { const int ti = 4 * K4 + l; L.tailpw[s * 8 + l] = ti < ns[s] ? tp[ns[s] + ti] : 0.0; }
      }
      tp += 2 * ns[s];
      for (int i = l; i <= ridx(LM ? cap[s] + EXT3 : cap[s]); i += NL) L.ring[s][i] = 0.0;
    }
    if constexpr (LM) {
      // this lane's run of chain positions (canon3_pack) and its static operands
      const bool dotl = l < 64 * D3;
      Canon3Lane cl;
      canon3_pack(ns, J3, D3, dotl ? 0 : 1, dotl ? l : l - 64 * D3, &cl, nullptr);
      const int st = cl.st, stride = dotl ? 8 : 4;
      const int nst = ns[st], K = nst >= 8 ? (dotl ? nst >> 3 : nst >> 2) : 0;
      const double *tps = tab;
      for (int s = 0; s < st; s++) tps += 2 * ns[s];
      int ro = 0;
      for (int s = 0; s < st; s++) ro += (int)(L.ring[s + 1] - L.ring[s]);
      q_ro[l] = ro; q_cap[l] = cap[st]; q_pos[l] = 0; q_sc[l] = st * 8 + cl.ch;
      q_tap0[l] = cl.act ? stride * (cl.m * J3) + cl.ch : 0;
      q_fl[l] = (cl.act && cl.m == 0 ? 1 : 0) | (cl.act && cl.m == cl.g - 1 ? 2 : 0) | (cl.act ? 4 : 0) | (cl.act ? 0 : 1);   // unassigned lanes start a (dead) chain of their own
      q_acc[l] = 0.0; q_mut[l] = 0.0; q_ttap[l] = 0;
      double *mtl = L.mt[0] + (size_t)((l >> 6) * J3) * 64 + (l & 63);
#pragma unroll
      for (int j = 0; j < J3; j++) {
        const int k = cl.m * J3 + j, tap = stride * k + cl.ch;
        const bool on = cl.act && k < K;
        if (dotl) { W[l].v[j] = 0.0; mtl[j * 64] = on ? tps[tap] : 0.0; }
        else W[l].v[j] = on ? tps[nst + tap] : 0.0;          // powtab; positions beyond the chain are empty
        XD[l].v[j] = 0.0;
      }
      if (dotl) {    // the chain's tail tap 8*K8 + c: owned by the chain's first lane
        const int tt = 8 * K + cl.ch;
        Wt[l].v[0] = 0.0;
        const bool on = cl.act && cl.m == 0 && tt < nst;
        q_mut[l] = on ? tps[tt] : 0.0;
        q_ttap[l] = tt < nst - 1 ? tt : nst - 1;
      }
    }
    if (l < 4) { L.bc[l] = 0.0; L.pv[l] = 0.0; }
    sa_stage_tables(L.libm, l, NL);
#pragma unroll
    for (int s = 0; s < 4; s++) if (l == s) { L.cst[s] = p.vmu[s]; L.cst[4 + s] = sum_powtab[s]; }
    if (l == 8) { L.cst[8] = p.proj_alpha; L.cst[9] = 1.0 - p.proj_alpha; L.cst[10] = p.mu_mix; L.cst[11] = p.mu_mix_beta; L.cst[12] = 1.0 - p.mu_mix_beta; L.cst[13] = p.lm_alpha; L.cst[14] = (double)p.lo; L.cst[15] = (double)p.hi; }
    if (l < 10) L.exwm[l] = 1.0 / 5;
    if (l < 16) L.hs[l] = (l == 12 || l == 13) ? 0.5 : 0.0;
    if (l < kRlsMax) L.rph[l] = 0.0;     // the wave-uniform state `us`: all zero at the start
    mr0[l] = (l >> 6) == 1 ? 1.0 / 5 : 0.0;   // wave 1: LS_ADA expert weights start at 1/5
    mr1[l] = 0.0; mr2[l] = 0.0; mr3[l] = 0.0;
    if (l >= 16 && l < 20) { mr2[l] = p.vmu[l - 16]; mr3[l] = sum_powtab[l - 16]; }
    if (l < kRlsMax * kRlsMax) L.P[l] = (l / kRlsMax == l % kRlsMax) ? 1.0 : 0.0;
  });
  ex.sync();
  const unsigned long long *exptab = reinterpret_cast<const unsigned long long *>(L.libm + 128 * 3);
  const double *tg[4] = {tab, tab + 2 * ns[0], tab + 2 * (ns[0] + ns[1]), tab + 2 * (ns[0] + ns[1] + ns[2])};   // per-stage {mutab, powtab} in global memory
  // CANON 2: lane-major copies (pred_tables.h): stage s = {mutab in dot order [RD*J][256], powtab in power-sum order [RD*J][256]}
  const double *tcm[4], *tcp[4];
  {
    const double *q_ = tabc;
#pragma unroll
    for (int s = 0; s < 4; s++) { tcm[s] = q_; tcp[s] = q_ + (CANON ? ROUNDS : 1) * C::slots(s) * 256; q_ += 2 * (CANON ? ROUNDS : 1) * C::slots(s) * 256; }
  }
  static_assert(CANON != 2 || (NL == kCanonNL && C::c0 == canon_slots(0) && C::c1 == canon_slots(1) && C::c2 == canon_slots(2) && C::c3 == canon_slots(3)), "lane-major tables are laid out for the (9,5,3,1) x 256 layout");
  // ring s starts ro1 + .. + ro_s doubles after ring[0]
  const int ro1 = (int)(L.ring[1] - L.ring[0]), ro2 = (int)(L.ring[2] - L.ring[1]), ro3 = (int)(L.ring[3] - L.ring[2]);
  const int rofs[4] = {0, ro1, ro1 + ro2, ro1 + ro2 + ro3};                                                    // search sweep: ring s = ring[0] + rofs[s] (one LDS base)
  // uniform mixer state (wave 0)
  // Wave-uniform state of the mixer chain that lives from one sample to the next -- ALC's S0, S1, the RLS phi / denom / 1/alpha (wave 2),
  // BlendExp's two loss EMAs (wave 3): seven doubles every lane of every wave held in registers although one wave uses each.  Round 6: they
  // live in LDS (the former rph array) and travel with loads / stores the waves issue anyway; the 14 registers were what spilled.
  double *const us = L.rph;       // [0] S0 [1] S1 [2] phi [3] denom [4] 1/alpha [5] [6] smrs
  bool have_prev = false;

  unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;   // optional section cycle counters (debug)
#define SA_TICK(i) do { if (prof) { const unsigned long long now_ = E::clock(); tp[i] += now_ - tc; tc = now_; } } while (0)
  if (prof) tc = E::clock();

  for (int t0 = 0; t0 < n; t0 += kLmsChunk) {
    // ---- stage a chunk of p_lpc / target in, flush the previous chunk of p_lpc+p_lms out
    if (!dec) ex.par([&](int l) {      // (decoder: inputs arrive and outputs leave sample by sample, see the head below)
      if (l < kLmsChunk) {
        if (t0 > 0) pout_g[t0 - kLmsChunk + l] = L.pout[l];
        if (t0 + l < n) { L.pin[l] = pin_g[t0 + l]; L.sv[l] = self[t0 + l]; }
      }
    });
    ex.sync();
    const int tend = (n - t0 < kLmsChunk) ? n - t0 : kLmsChunk;
    for (int tt = 0; tt < tend; tt++) {
      const int par = tt & 1;
      // ---- A: fused sweep (update of previous step, predict of this step)
      if constexpr (!CANON) {
      ex.par([&](int l) { lms_sweep<NL, C, lms_ahead<C>()>(ex, W[l], MT[l], PT[l], acc[l], L.ring[0], rofs, pos, L.bc, mjv, l); });
      SA_TICK(0);
      ex.wave_sum8x(acc);
      SA_TICK(1);
      ex.par([&](int l) {   // lane 16r of each wave holds the wave totals of values 2r, 2r+1
        if ((l & 15) == 0) {
          double *dst = L.part + (par * NW + (l >> 6)) * 8 + 2 * ((l >> 4) & 3);
          dst[0] = acc[l].v[0]; dst[1] = acc[l].v[1];
        }
      });
      } else if constexpr (LM) {
      // Lane-map layout: one parallel phase (history loads, weight update of the dot lanes, squares of the power lanes),
      // then all chains hop along their lane runs at once.
      ex.par([&](int l) {
        const bool dotl = l < 64 * D3;
        int u0 = q_pos[l] + q_tap0[l]; if (u0 >= q_cap[l]) u0 -= q_cap[l];
        const double *rg = L.ring[0] + q_ro[l];
        if (dotl) {
          // positions k0 + j -> taps 8 (k0 + j) + c: padded ring index ridx(u0 + 8j) = ridx(u0) + 9j (no wrap: mirrored ring)
          const int bn = ridx(u0), bo = ridx(u0 + 1);
          const int st = q_sc[l] >> 3;
          const double wg = L.bc[st];
          const double *mtl = L.mt[0] + (size_t)((l >> 6) * J3) * 64 + (l & 63);
          // Round 6: written as a software pipeline (history / mutab loads of position j + 3 in flight under position j's update,
          // a scheduling fence per position).  Left to itself the scheduler issues all 3 J loads of the lane first -- 6 J transient
          // registers on top of the 4 J that W and XD need anyway: 108 registers of the 17-position layout, 148 of the 33-position
          // one and 288 B of the 49-position one went to scratch, reloaded in the middle of every sample.
          constexpr int GA = 3 < J3 ? 3 : J3;
          double bo_[GA], bm_[GA];
          auto ldj = [&](auto JC) {
            constexpr int j = decltype(JC)::value;
            XD[l].v[j] = rg[bn + 9 * j]; bo_[j % GA] = rg[bo + 9 * j]; bm_[j % GA] = mtl[j * 64];
          };
          static_for<0, GA>(ldj);
          static_for<0, J3>([&](auto JC) {
            constexpr int j = decltype(JC)::value;
            const double xo = bo_[j % GA], mt = bm_[j % GA];
            if constexpr (j + GA < J3) ldj(std::integral_constant<int, j + GA>{});
            double w = fma(mt, wg * xo, W[l].v[j]);       // mutab is 0 at empty positions: the weight stays 0
            w = clampd(w, -10.0, 10.0);
            SA_PIN_F64(w);
            W[l].v[j] = w;
            SA_SCHED_FENCE();
          });
          {   // the chain's tail tap (first lane of the chain; elsewhere its mutab is 0)
            int in = q_pos[l] + q_ttap[l]; if (in >= q_cap[l]) in -= q_cap[l];
            const double xo = rg[ridx(in + 1)];
            double w = fma(q_mut[l], wg * xo, Wt[l].v[0]);
            w = clampd(w, -10.0, 10.0);
            Wt[l].v[0] = w;
            if ((q_fl[l] & 5) == 5) L.tailw[(par * 4 + st) * 8 + (q_sc[l] & 7)] = w;
          }
        } else {
          // taps 4 (k0 + j) + c4: even j at ridx(u0) + 9 (j/2), odd j at ridx(u0 + 4) + 9 (j/2)
          const int be = ridx(u0), bd = ridx(u0 + 4);
          constexpr int GA = 4 < J3 ? 4 : J3;                 // same pipeline: the square of position j under the loads of j + 1 .. j + 4
          double bx_[GA];
          auto ldj = [&](auto JC) { constexpr int j = decltype(JC)::value; bx_[j % GA] = rg[((j & 1) ? bd : be) + 9 * (j >> 1)]; };
          static_for<0, GA>(ldj);
          static_for<0, J3>([&](auto JC) {
            constexpr int j = decltype(JC)::value;
            const double xs = bx_[j % GA];
            if constexpr (j + GA < J3) ldj(std::integral_constant<int, j + GA>{});
            double q = xs * xs;
            SA_PIN_F64(q);
            XD[l].v[j] = q;
            SA_SCHED_FENCE();
          });
        }
        q_acc[l] = 0.0;
      });
      SA_TICK(0);
      {
        const int G = ex.wave_hops(64 * D3, G3d, G3p);
        for (int h = 0; h < G; h += 2) {
#pragma unroll
          for (int u = 0; u < 2; u++) {
            typename E::template Reg<double> sh = q_acc;
            ex.shift_up1(sh);
            ex.par([&](int l) {
              double a = (q_fl[l] & 1) ? 0.0 : sh[l];
#pragma unroll
              for (int j = 0; j < J3; j++) a = fma(XD[l].v[j], W[l].v[j], a);
              q_acc[l] = a;
            });
          }
        }
      }
      ex.par([&](int l) {
        if ((q_fl[l] & 6) == 6) {
          if (l < 64 * D3) L.csum[(par * 4 + (q_sc[l] >> 3)) * 8 + (q_sc[l] & 7)] = q_acc[l];
          else L.psum[(par * 4 + (q_sc[l] >> 3)) * 4 + (q_sc[l] & 7)] = q_acc[l];
        }
      });
      SA_TICK(1);
      } else {
      // Canonical order (slmath::dot / calc_s2pow): weight update of this lane's chain positions, then the
      // running sums hop along the lanes of each chain.  After hop h the sum held by lane m <= h is final, so
      // after H = ceil(K / J) hops lane H-1 holds the chain total.
      // Stage by stage: power-sum chains (history squares + powtab in registers), then weight update and dot chains
      // (history values in registers).  Only one stage's chain operands are live at a time, and the loops between
      // the phases keep the compiler from hoisting every LDS load of the sample to the top (which spilled).
      // One tight loop per chain set, four hops per iteration: a taken branch costs about as much as four
      // dependent FMAs, and hops beyond H change nothing (the sum of lane m is final from hop m on).
      auto canon_stage = [&](auto SC) {
        constexpr int s = decltype(SC)::value;     // compile-time stage index: every register array below is indexed statically
        constexpr int f = C::first(s);
        const double *ring = L.ring[s];
        const int last = ns[s] - 1, cp = cap[s], ps = pos[s];
        const int K8 = ns[s] >= 8 ? ns[s] >> 3 : 0, K4 = ns[s] >= 8 ? ns[s] >> 2 : 0;
        // power-sum chains: chain c4 runs over the 64 lanes of wave c4, J positions per lane; with 512 lanes the chain
        // makes SMUL = 2 rounds over the wave (positions r*64*J + lane*J + j), so that only J operand pairs are live
        ex.par([&](int l) { sq[s][l] = 0.0; });
        static_for<0, RP>([&](auto RC) {
          constexpr int r = decltype(RC)::value;
          const int kbase = r * 64 * C::slots(s);
          if (kbase < K4 || r == 0) {
            const int Hr = (K4 - kbase + C::slots(s) - 1) / C::slots(s) < 64 ? (K4 - kbase + C::slots(s) - 1) / C::slots(s) : 64;
            ex.par([&](int l) {
              const int lw = l & 63, c4 = l >> 6;
              if (c4 < 4) {
#pragma unroll
                for (int j = 0; j < C::slots(s); j++) {
                  const int k = kbase + lw * C::slots(s) + j;
                  int tap = 4 * k + c4; tap = tap < last ? tap : last;
                  int in = ps + tap; if (in >= cp) in -= cp;
                  const double xs = ring[ridx(in)];
                  XX[l].v[j] = xs * xs;
                  const double pw = CANON == 2 ? tcp[s][(r * C::slots(s) + j) * NL + l] : L.pt[s][ridx(tap)];
                  PR[l].v[j] = k < K4 ? pw : 0.0;                                        // positions >= K4 are empty
                }
              }
              if (r > 0) hop[l] = ex.wave_lane(sq[s], l, 63);     // the chain re-enters lane 0 with what lane 63 ended the last round with
            });
            for (int h = 0; h < Hr; h += 4) {
#pragma unroll
              for (int u = 0; u < 4; u++) {
                typename E::template Reg<double> sh = sq[s];
                ex.shift_up1(sh);
                ex.par([&](int l) {
                  double a = (l & 63) == 0 ? (r > 0 ? hop[l] : 0.0) : sh[l];
#pragma unroll
                  for (int j = 0; j < C::slots(s); j++) a = fma(PR[l].v[j], XX[l].v[j], a);
                  sq[s][l] = a;
                });
              }
            }
            ex.par([&](int l) {
              const int lw = l & 63, c4 = l >> 6;
              if (Hr > 0 && c4 < 4 && lw == Hr - 1) L.psum[(par * 4 + s) * 4 + c4] = sq[s][l];
            });
          }
        });
        // dot chains: chain c runs over LPC lanes (CPW chains per wave), RD rounds; the weights are updated on the way
        ex.par([&](int l) { sd[s][l] = 0.0; });
        static_for<0, RD>([&](auto RC) {
          constexpr int r = decltype(RC)::value;
          const int kbase = r * LPC * C::slots(s);
          if (kbase < K8 || r == 0) {
            const int Hr = (K8 - kbase + C::slots(s) - 1) / C::slots(s) < LPC ? (K8 - kbase + C::slots(s) - 1) / C::slots(s) : LPC;
            ex.par([&](int l) {
              const int lw = l & 63, c = (l >> 6) * CPW + lw / LPC, m = lw % LPC;
              const double wg = L.bc[s];
#pragma unroll
              for (int j = 0; j < C::slots(s); j++) {
                const int k = kbase + m * C::slots(s) + j;
                int tap = 8 * k + c; tap = tap < last ? tap : last;
                int in = ps + tap; if (in >= cp) in -= cp;
                const double xn = ring[ridx(in)], xo = ring[ridx(in + 1)];
                const double mu_t = k < K8 ? (CANON == 2 ? tcm[s][(r * C::slots(s) + j) * NL + l] : L.mt[s][ridx(tap)]) : 0.0;   // positions >= K8 are empty: the weight stays 0
                double w = fma(mu_t, wg * xo, W[l].v[f * RD + r * C::slots(s) + j]);
                w = clampd(w, -10.0, 10.0);
                W[l].v[f * RD + r * C::slots(s) + j] = w;
                XD[l].v[j] = xn;
              }
              if (r == 0) {   // the chain's tail tap (lanes m == 0; elsewhere mutab is 0 and the weight stays 0)
                int tap = 8 * K8 + c; tap = tap < last ? tap : last;
                int in = ps + tap; if (in >= cp) in -= cp;
                const double xo = ring[ridx(in + 1)];
                const double mu_t = (m == 0 && 8 * K8 + c < ns[s]) ? (CANON == 2 ? tg[s][tap] : L.mt[s][ridx(tap)]) : 0.0;
                double w = fma(mu_t, wg * xo, Wt[l].v[s]);
                w = clampd(w, -10.0, 10.0);
                Wt[l].v[s] = w;
                if (m == 0) L.tailw[(par * 4 + s) * 8 + c] = w;
              }
              if (r > 0) {   // re-enter lane 0 of the chain with its last lane's sum (uniform lane index per read)
#pragma unroll
                for (int q = 0; q < CPW; q++) { const double t_ = ex.wave_lane(sd[s], l, q * LPC + LPC - 1); if (lw / LPC == q) hop[l] = t_; }
              }
            });
            for (int h = 0; h < Hr; h += 4) {
#pragma unroll
              for (int u = 0; u < 4; u++) {
                typename E::template Reg<double> sh = sd[s];
                ex.shift_up1(sh);
                ex.par([&](int l) {
                  double a = ((l & 63) % LPC) == 0 ? (r > 0 ? hop[l] : 0.0) : sh[l];
#pragma unroll
                  for (int j = 0; j < C::slots(s); j++) a = fma(XD[l].v[j], W[l].v[f * RD + r * C::slots(s) + j], a);
                  sd[s][l] = a;
                });
              }
            }
            ex.par([&](int l) {
              const int lw = l & 63, c = (l >> 6) * CPW + lw / LPC, m = lw % LPC;
              if (Hr > 0 && m == Hr - 1) L.csum[(par * 4 + s) * 8 + c] = sd[s][l];
            });
          }
        });
      };
      canon_stage(std::integral_constant<int, 0>{}); canon_stage(std::integral_constant<int, 1>{});
      canon_stage(std::integral_constant<int, 2>{}); canon_stage(std::integral_constant<int, 3>{});
      SA_TICK(0);
      SA_TICK(1);
      }
      ex.sync();
      SA_TICK(2);
      // ---- B: mixer chain.  Wave 0 turns the stage sums into the prediction (head), then into the
      // stage gains.  The three updates that only feed the NEXT prediction run beside the stage gains
      // on waves 1..3, between barrier H (head published) and barrier 2:
      //   wave 1  LS_ADA expert weights            wave 2  RLS / ALC            wave 3  BlendExp softmax
      // The RLS P-matrix update is deferred until after the next sweep's barrier, where it hides
      // under wave 0's head.
      // Round 6.  Per-wave section counters showed that THIS is the sample's critical path, not wave 0's head: wave 2 needed 2 100
      // cycles from the sweep's barrier to barrier H (wave 0: 1 500) and 2 000 more behind it, nearly all of it LDS round trips -- P row
      // load / store / reload, the RLS history, P x and the weights each travelled through LDS between lanes that sit in ONE wave.  Now lane
      // l of wave 2 keeps x[l], w[l] and (P x)[l] in registers, what a dot product needs from the other lanes comes by v_readlane (uniform
      // values, scalar registers), the history rolls by one DPP shift, and only the lane's own P row is read and written in LDS, once.
      // Same operations in the same order (dot_canon_n<M> == dot_canon for M terms): bit-identical.
      rls_dispatch(m, [&](auto MC) {
        constexpr int M = decltype(MC)::value;
        double xu[M], phu[M];
        ex.wave(2, [&]() {
#pragma unroll
          for (int j = 0; j < M; j++) { xu[j] = ex.lane_bcast(x_r, 128 + j); phu[j] = ex.lane_bcast(ph_r, 128 + j); }    // RLS history; P x of the PREVIOUS step
        });
        ex.wave_par(2, [&](int g) {
          const int l = g & 63;
          if (l < M) {
            double *prow = L.P + l * kRlsMax;
            double pr[M];
#pragma unroll
            for (int j = 0; j < M; j++) pr[j] = prow[j];
            if (have_prev) {            // P update of row l (rls.cpp:47-56) of the PREVIOUS step, deferred to here where it hides under wave 0's head
              const double phl = ph_r[g], denom = us[3], inv_alpha = us[4];
#pragma unroll
              for (int j = 0; j < M; j++) { pr[j] = fma(-denom, phl * phu[j], pr[j]) * inv_alpha; prow[j] = pr[j]; }
            }
            // ph = P x, row l (rls.cpp:33): needs only P and the RLS history, both final by now
            ph_r[g] = dot_canon_n<M>([&](int j) { return pr[j]; }, [&](int j) { return xu[j]; });
          }
        });
        ex.wave(2, [&]() {
          double p2[M];
#pragma unroll
          for (int j = 0; j < M; j++) p2[j] = ex.lane_bcast(ph_r, 128 + j);
          const double phi = fmax(dot_canon_n<M>([&](int j) { return xu[j]; }, [&](int j) { return p2[j]; }), 1e-8);
          if (ex.is_lane0w()) us[2] = phi;
        });
      });
      // Wave 0 (round 6): every LDS input of the head -- last step's blend weights, RLS prediction and expert weights, the constants, this
      // sample's p_lpc and value -- is requested HERE, together with the stage totals' loads: one LDS round trip for the whole head where
      // the compiler's placement (loads next to their first use, under register pressure) had eight in a row.
      double h_smw0 = 0.0, h_smw1 = 0.0, h_rpx = 0.0, h_plpc = 0.0;
      int h_sv = 0;
      ex.wave(0, [&]() {
        h_smw0 = L.hs[12]; h_smw1 = L.hs[13]; h_rpx = L.hs[10];
        if (!dec) { h_plpc = L.pin[tt]; h_sv = L.sv[tt]; }
      });
      ex.wave_par(0, [&](int l) {
        if constexpr (!CANON) {
          if (l >= 16 && l < 20) {   // cross-wave totals of stage l-16, waves in order (lanes 16..19 own the stage gains)
            const int s = l - 16;
            double pa_[NW], pb_[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) { pa_[w] = L.part[(par * NW + w) * 8 + s]; pb_[w] = L.part[(par * NW + w) * 8 + 4 + s]; }
            SA_SCHED_FENCE();                  // all partial sums requested before the first addition
            double a = pa_[0], b = pb_[0];
#pragma unroll
            for (int w = 1; w < NW; w++) { a = a + pa_[w]; b = b + pb_[w]; }
            dots_r[l] = a; spow_r[l] = b;      // (the head takes the stage predictions from these lanes by v_readlane: no LDS hand-over)
          }
        } else {
          // slmath::dot: sum1 + sum2 lane-wise, ((b0+b1)+b2)+b3, += transform_reduce tail; calc_s2pow alike (common/math.h:130-191).
          // Round 6: EIGHT lanes, one instruction stream -- lane 16 + s forms the dot total of stage s, lane 20 + s its power sum.  The two
          // have the same shape once the operands are named A, B: dot A = x, B = w; power sum A = x x, B = powtab; chains q[c] + q[c + 4]
          // with -0.0 in the power sum's upper four (x + -0.0 = x for every x); so both run side by side where lanes 16..19 used to do
          // one after the other.  All 22 operands of a lane are requested first (tap indices clamped to the stage's last tap, the tail
          // weight slots always written: unused operands are finite, the selects below discard them), then the arithmetic behind a
          // scheduling fence.  The power sum reaches the gains lane 16 + s through pv[s] (read behind the head barrier).
          if (l >= 16 && l < 24) {
            const int s = (l - 16) & 3;
            const bool pw = l >= 20;
            const int nsl = s == 0 ? ns[0] : (s == 1 ? ns[1] : (s == 2 ? ns[2] : ns[3]));
            const int cs = s == 0 ? cap[0] : (s == 1 ? cap[1] : (s == 2 ? cap[2] : cap[3]));
            const int ps = s == 0 ? pos[0] : (s == 1 ? pos[1] : (s == 2 ? pos[2] : pos[3]));
            const int ro = (s >= 1 ? ro1 : 0) + (s >= 2 ? ro2 : 0) + (s >= 3 ? ro3 : 0);   // sums, not a select among captured variables (which would pin the closure, and everything it references, in scratch)
            const double *r0 = L.ring[0];
            const int K8 = nsl >= 8 ? nsl >> 3 : 0, K4 = nsl >= 8 ? nsl >> 2 : 0;
            const int last = nsl - 1;
            const int t0 = pw ? 4 * K4 : 8 * K8;              // first tail tap
            const int rl = nsl - t0;                          // tail length: < 8 (dot) resp. < 4 (power sum), or the whole stage when n < 8
            auto hist = [&](int tap) { tap = tap < last ? tap : last; int in = ps + tap; if (in >= cs) in -= cs; return r0[ro + ridx(in)]; };
            const double *qs = pw ? L.psum + (par * 4 + s) * 4 : L.csum + (par * 4 + s) * 8;     // (one base: both arrays lie in the same LDS block, psum behind csum)
            const double *bs = pw ? L.tailpw + s * 8 : L.tailw + (par * 4 + s) * 8;
            double q[8], xa[7], bb[7];
#pragma unroll
            for (int u = 0; u < 8; u++) q[u] = qs[u];
#pragma unroll
            for (int u = 0; u < 7; u++) { xa[u] = hist(t0 + u); bb[u] = bs[u]; }
            SA_SCHED_FENCE();
#pragma unroll
            for (int u = 4; u < 8; u++) q[u] = pw ? -0.0 : q[u];
#pragma unroll
            for (int u = 0; u < 7; u++) xa[u] = xa[u] * (pw ? xa[u] : 1.0);      // A: x (x * 1.0 = x exactly) resp. x x
            double tot = 0.0;
            if (K8 > 0) {
              const double q0 = q[0] + q[4], q1 = q[1] + q[5], q2 = q[2] + q[6], q3 = q[3] + q[7];
              tot = ((q0 + q1) + q2) + q3;
            }
            const bool g4 = rl >= 4;
            const double v1 = fma(xa[1], bb[1], xa[0] * bb[0]), v2 = fma(xa[3], bb[3], xa[2] * bb[2]);
            double init = g4 ? 0.0 + (v1 + v2) : 0.0;
            const int rem = g4 ? rl - 4 : rl;
            const double e0 = g4 ? xa[4] : xa[0], e1 = g4 ? xa[5] : xa[1], e2 = g4 ? xa[6] : xa[2];
            const double f0 = g4 ? bb[4] : bb[0], f1 = g4 ? bb[5] : bb[1], f2 = g4 ? bb[6] : bb[2];
            const double i2 = (init + e0 * f0) + e1 * f1;
            init = rem >= 2 ? i2 : init;
            const double el = rem >= 2 ? e2 : e0, fl = rem >= 2 ? f2 : f0;
            const double i3 = fma(el, fl, init);
            init = (rem & 1) ? i3 : init;
            tot = tot + init;
            if (pw) L.pv[s] = tot; else dots_r[l] = tot;
          }
        }
      });
#if defined(SACAMD_EXP_TICK_TOTALS)
      SA_TICK(7);                            // build-time probe: section 7 ("bar2" column) = the stage totals, "head" = what follows them
#endif
      double bp[5] = {0, 0, 0, 0, 0};
      bool dec_ok = true;
      ex.wave(0, [&]() {
        const double smw0 = h_smw0, smw1 = h_smw1;
        const double pa = L.cst[8], pa1 = L.cst[9], lo = L.cst[14], hi = L.cst[15];        // proj_alpha, 1 - proj_alpha, Cascade clamp range (with the expert weights: one batch)
        // Cascade::Predict (cascade.h:93-100)
        const double rpx = h_rpx;                          // dot(rx, rw), left in hs[10] by wave 2 after its update
        double pl[5], ep[2], exw[10];      // (the ten expert weights are loaded here, not with the early batch: 20 registers live across the totals cost more in spills than the round trip saves)
#pragma unroll
        for (int i = 0; i < 10; i++) exw[i] = L.exwm[i];
#pragma unroll
        for (int i = 0; i < 4; i++) pl[i] = ex.lane_bcast(dots_r, 16 + i);
        pl[4] = rpx;
        for (int e = 0; e < 2; e++) ep[e] = dot_canon_n<5>([&](int i) { return pl[i]; }, [&](int i) { return exw[5 * e + i]; });
        const double pred = dot_canon_n<2>([&](int i) { return i ? ep[1] : ep[0]; }, [&](int i) { return i ? smw1 : smw0; });
        // The OLS prediction joins here; then the target of all updates, val - p_lpc (pred.cpp:43).  Decoder: p_lpc comes from
        // the OLS kernel of this channel running beside this one, the sum goes to the bias kernel, whose decoded sample comes back.
        double plpc, target;
        if (!dec) {
          plpc = h_plpc;
          L.pout[tt] = plpc + pred;
          target = (double)h_sv - plpc;
        } else {
          const int t = t0 + tt;
          dec_ok = sa_wait_ge(dec->prog_in, t + 1, dec->fail);
          plpc = pin_g[t];
          if (ex.is_lane0()) { pout_g[t] = plpc + pred; sa_publish(dec->prog_out, t + 1); }
          dec_ok = dec_ok && sa_wait_ge(dec->prog_self, t + 1, dec->fail);
          target = (double)self[t] - plpc;
          if (!dec_ok && ex.is_lane0()) L.hs[15] = 1.0;       // a partner kernel is not there: everybody leaves after the barrier
        }
        // Cascade::Update(target): stage targets (cascade.h:101-112)
        double p_prefix = 0.0;
#pragma unroll
        for (int i = 0; i <= 4; i++) {
          const double ew0 = exw[i], ew1 = exw[5 + i];
          const double wgt = fmax(dot_canon_n<2>([&](int q) { return q ? ew1 : ew0; }, [&](int q) { return q ? smw1 : smw0; }), 0.0);
          const double px = fma(pa1, p_prefix, pa * pred);
          bp[i] = target - clampd(px, lo, hi);
          p_prefix = fma(wgt, pl[i], p_prefix);
        }
        if (ex.is_lane0()) {   // hand-over to the update waves
          L.hs[0] = target; L.hs[1] = ep[0]; L.hs[2] = ep[1];
          for (int i = 0; i < 5; i++) L.hs[3 + i] = pl[i];
          L.hs[8] = bp[4]; L.hs[9] = rpx;
        }
      });
      SA_TICK(3);
      // Barrier H: the head has published this step's prediction pieces.  Wave 0 goes on to the stage
      // gains (all the next sweep needs from it); waves 1..3 run, beside it, the three updates that only
      // feed the NEXT prediction, so that after barrier 2 all four waves start the next sweep together.
      ex.sync();
      if (dec && L.hs[15] != 0.0) return;
      // ---- wave 1: LS_ADA experts (ls.h:224-236), lanes 0..9: expert e = l/5 (0: L1 loss, 1: L2), input i = l%5
      ex.wave_par(1, [&](int g) {
        const int l = g & 63;
        if (l < 10) {
          const int e = l >= 5, i = l - 5 * e;
          const double error = L.hs[0] - L.hs[1 + e];
          const double loss = e ? error : sgnd(error);
          const double grad = loss * L.hs[3 + i];
          const double beta = L.cst[11], beta1 = L.cst[12];
          exeg_r[g] = fma(beta, exeg_r[g], beta1 * grad * grad);
          const double mu_scaled = L.cst[10] / (sqrt(exeg_r[g]) + 1e-5);
          exw_r[g] = fma(mu_scaled, grad, exw_r[g]);
          L.exwm[l] = exw_r[g];
        }
      });
      // ---- wave 2: RLS::Update + ALC (rls.cpp:28-56, rls.h:21-39) except the P update (deferred);
      // ph = P x and phi were computed before the barrier (they do not depend on this step's prediction)
      double rerr = 0.0, alpha = 0.0, rbp4 = 0.0, phi = 0.0, denom = 0.0;
      ex.wave(2, [&]() {
        rbp4 = L.hs[8];
        const double rpx_ = L.hs[9], lm_alpha = L.cst[13], S0 = us[0], S1 = us[1];
        phi = us[2];
        SA_SCHED_FENCE();                    // the six loads leave together (the shape parameter used to be fetched after the division: one more LDS round trip in the chain)
        rerr = rbp4 - rpx_;
        const double err2 = rerr * rerr;
        const double R = fmax(S0 - S1, 1e-5);
        const double nis = err2 / (phi + R);
        const double mm = sa_exp_t(-lm_alpha * nis, exptab);
        alpha = fma(0.999 - 0.99, mm, 0.99);
        if (ex.is_lane0w()) { us[0] = fma(0.95, S0, (1.0 - 0.95) * err2); us[1] = fma(0.95, S1, (1.0 - 0.95) * phi); }
      });
      ex.wave_par(2, [&](int g) {            // both reciprocals in one instruction stream: lane 0 denom = 1 / (alpha + phi), lane 1 1 / alpha (for the next step's P update)
        if ((g & 63) < 2) { rcp_r[g] = 1.0 / ((g & 63) == 0 ? alpha + phi : alpha); us[3 + (g & 63)] = rcp_r[g]; }
      });
      ex.wave(2, [&]() { denom = ex.lane_bcast(rcp_r, 128); });
      ex.wave_par(2, [&](int g) { if ((g & 63) < m) rw_r[g] = fma(rerr, denom * ph_r[g], rw_r[g]); });
      ex.wave_shift_up1(2, x_r);                                                              // RollBack(x, val), rls.cpp:64: x[l] <- x[l-1] ...
      ex.wave_par(2, [&](int g) { if ((g & 63) == 0) x_r[g] = rbp4; });                       // ... and x[0] <- val (the RLS stage's target bp[4])
      rls_dispatch(m, [&](auto MC) {     // RLS::Predict of the NEXT step (rls.cpp:21-26): its inputs are final now
        constexpr int M = decltype(MC)::value;
        ex.wave(2, [&]() {
          double xu[M], wu[M];
#pragma unroll
          for (int j = 0; j < M; j++) { xu[j] = ex.lane_bcast(x_r, 128 + j); wu[j] = ex.lane_bcast(rw_r, 128 + j); }
          const double rpx_next = dot_canon_n<M>([&](int j) { return xu[j]; }, [&](int j) { return wu[j]; });
          if (ex.is_lane0w()) L.hs[10] = rpx_next;
        });
      });
      // ---- wave 3: BlendExp<RunSumEMA>::Update (blend.h:31-90)
      double zm[2] = {0, 0}, maxz = 0.0;
      ex.wave(3, [&]() {
        const double tg_ = L.hs[0], e0_ = L.hs[1], e1_ = L.hs[2], sm0 = us[5], sm1 = us[6];
        SA_SCHED_FENCE();
        for (int e = 0; e < 2; e++) {
          const double loss = fabs(tg_ - (e ? e1_ : e0_));
          const double sm = fma(0.95, e ? sm1 : sm0, (1.0 - 0.95) * (-loss));
          if (ex.is_lane0w()) us[5 + e] = sm;
          zm[e] = 1.0 * sm;
        }
        maxz = fmax(zm[0], zm[1]);
      });
      ex.wave_par(3, [&](int g) { if ((g & 63) < 2) exz_r[g] = sa_exp_t(((g & 63) == 0 ? zm[0] : zm[1]) - maxz, exptab); });
      ex.wave(3, [&]() {
        const double w0 = ex.lane_bcast(exz_r, 192), w1 = ex.lane_bcast(exz_r, 193);
        const double inv = 1.0 / (w0 + w1);
        if (ex.is_lane0w()) { L.hs[12] = w0 * inv; L.hs[13] = w1 * inv; }
      });
      ex.wave_par(0, [&](int l) {
        // lanes 16..19: NLMS_Stream::Update scalar part (ls.h:47-48) + history push of stage l-16
        if (l >= 16 && l < 20) {
          const int sl = l - 16;
          const double bps = sl == 0 ? bp[0] : (sl == 1 ? bp[1] : (sl == 2 ? bp[2] : bp[3]));
          const int ps = sl == 0 ? pos[0] : (sl == 1 ? pos[1] : (sl == 2 ? pos[2] : pos[3]));
          const int cs = sl == 0 ? cap[0] : (sl == 1 ? cap[1] : (sl == 2 ? cap[2] : cap[3]));
          // one LDS base + offset: a select among the four ring pointers becomes a load through a selected ADDRESS
          // inside the LmsLds object, which pins that object (and every pointer in it) in scratch memory
          double *rg = L.ring[0] + ((sl >= 1 ? ro1 : 0) + (sl >= 2 ? ro2 : 0) + (sl >= 3 ? ro3 : 0));
          const double spow = CANON ? L.pv[sl] : spow_r[l];       // canonical layouts: the power sum was formed by lane 20 + sl (totals above)
          L.bc[sl] = vmu_r[l] * (bps - dots_r[l]) * spt_r[l] / (spow + 1.0);
          int np = ps - 1; if (np < 0) np += cs;
          rg[ridx(np)] = bps;
          if (LM ? np < EXT3 : np == 0) rg[ridx(cs + np)] = bps;        // mirror: ring[in + 1] (lane-map layout: a lane's whole window) needs no wrap in the sweep
          if constexpr (!CANON && kLmsMtFac) {
            // factored step sizes: everything behind the window stays zero -- the value that was the oldest one the update read (offset
            // nsl from the old position = nsl + 1 from the new one) leaves the window now
            const int nsl = sl == 0 ? ns[0] : (sl == 1 ? ns[1] : (sl == 2 ? ns[2] : ns[3]));
            if (nsl + 1 < cs) {
              int zi = np + nsl + 1; if (zi >= cs) zi -= cs;
              double zero = 0.0;
              SA_PIN_F64(zero);              // materialised here (hoisted out of the sample loop the constant was spilled to scratch and reloaded per sample)
              rg[zi] = zero;
              if (zi == 0) rg[cs] = zero;
            }
          }
        }
      });
      SA_TICK(4);
      ex.sync();
      SA_TICK(5);
      have_prev = true;
      #pragma unroll
      for (int s = 0; s < 4; s++) { pos[s] -= 1; if (pos[s] < 0) pos[s] += cap[s]; }
      if constexpr (LM) ex.par([&](int l) { int q = q_pos[l] - 1; if (q < 0) q += q_cap[l]; q_pos[l] = q; });
      SA_TICK(6);
      SA_TICK(7);
    }
  }
#ifndef SACAMD_EXP_PROF_WAVE
#define SACAMD_EXP_PROF_WAVE 0          // build-time knob (tools/build_variant.sh): the wave whose section counters are reported (every wave keeps its own)
#endif
  if (prof) ex.par([&](int l) { if (l == 64 * SACAMD_EXP_PROF_WAVE) for (int i = 0; i < 8; i++) prof[8 + i] = tp[i]; });
#undef SA_TICK
  // flush the last chunk
  if (!dec) ex.par([&](int l) {
    const int t0 = ((n - 1) / kLmsChunk) * kLmsChunk;
    if (n > 0 && l < kLmsChunk && t0 + l < n) pout_g[t0 + l] = L.pout[l];
  });
}

}  // namespace sacamd
