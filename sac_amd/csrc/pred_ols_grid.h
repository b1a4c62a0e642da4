// sac_amd/csrc/pred_ols_grid.h -- stage 1 (OLS) for regressors of 33..64 taps on ONE wave, matrix spread 2D-cyclically
// over the lanes (round 5).
//
// Reference: OLS (/root/reference/src/pred/ols.cpp:7-57) + slmath::LDLT (common/math.h:14-78) + RunSumGEO
// (common/utils.h:39-72), regressor of Predictor::fillbuf_ch0/ch1 (libsac/pred.cpp:17-31) -- the arithmetic of
// ols_stage_reg (pred_ols.h), element for element and in the same order, so p_lpc stays bit-identical.
//
// Why.  ols_stage_reg gives lane l the matrix ROW l: the rank-one update of column step k costs 3 (n - k) wave instructions
// of which lane l uses l - k, i.e. a third of the issued lane-operations on average (the matrix is a triangle, the lanes run
// in lock step), and the stage is bound by instruction issue.  Here lane l = (a, b) = (l & 7, l >> 3) holds the elements
// (i, j) with i = a, j = b (mod 8): the lower triangle of 8 x 8 blocks (I, J), I >= J, is NB (NB + 1) / 2 register slots per
// lane, all 64 lanes are busy in every slot, and a column step K = k / 8 only touches the slots with J >= K:
//     sum_K 8 (NB - K)(NB - K + 1) / 2  slot updates   (NB = 8: 960 against 2016 column-slots of the row layout).
// Every element still receives its terms in ascending k with the reference's fused / unfused pattern (canon.h fold_add), so
// the result is bit-identical.  Dead elements (row or column <= k, or >= n) are updated with whatever their lanes read --
// nobody reads them again -- so the kernel needs neither masks nor zero padding.
//
// L and z live in LDS as ONE linear stream U in the order the back-substitution (math.h:67-73) consumes them: with
// ip = n-1-i and kp = n-1-k, row ip occupies U[ip (ip+1) / 2 + 0 .. ip]: z[i] first, then L[k][i] for k = i+1 .. n-1.  The
// backward solve -- n^2 / 2 dependent FMAs on one lane, the floor any implementation of that summation order has -- is then a
// single software-pipelined pass over U (loads several chunks ahead, no per-row start-up), and column k of the factorisation
// is the contiguous run U[ck - k + r], r = k+1 .. n-1, with ck = ipk (ipk + 1) / 2, ipk = n-1-k.
#pragma once
#include <type_traits>
#include "pred_ols.h"

namespace sacamd {

SA_HD constexpr int grid_slot(int I, int J) { return I * (I + 1) / 2 + J; }
SA_HD constexpr int grid_tri1(int ip) { return ip * (ip + 1) / 2; }
// row (ip) of stream position t
SA_HD constexpr int grid_row_of(int t) { int ip = 0; while (grid_tri1(ip + 1) <= t) ip++; return ip; }

// f(integral_constant<I>) for I = A .. B-1 until one returns true
template <int A, int B, class F>
SA_HD __attribute__((always_inline)) bool grid_static_any(F &&f) {
  if constexpr (A < B) {
    if (f(std::integral_constant<int, A>{})) return true;
    return grid_static_any<A + 1, B>(f);
  } else {
    return false;
  }
}

constexpr int kGridGuard = 72;       // doubles in front of / behind U that stray (dead-element) reads may touch
struct OlsLdsGrid {
  double *X, *Wv, *dump, *libm, *U;
  SA_HD static size_t bytes(int nmax) {
    return (size_t)(nmax + (nmax + kOlsPad) + 64 + kLibmLdsDoubles + kGridGuard + grid_tri1(nmax) + kGridGuard) * sizeof(double) + 16;
  }
  SA_HD void carve(char *base, int nmax) {
    double *d = reinterpret_cast<double *>(base);
    X = d; d += nmax; Wv = d; d += nmax + kOlsPad; dump = d; d += 64; libm = d; d += kLibmLdsDoubles;
    d += kGridGuard; U = d;
  }
};

template <int N> struct OlsGridRegs { double v[N]; };

// ---- backward substitution: one pass over the stream.  Sixteen consecutive stream values are ONE register (lane l holds
// U[16 c + (l & 15)], every row of 16 lanes the same sixteen), and a term of the chain is ONE instruction,
//     v_fmac_f64_dpp s, chunk row_newbcast:q, w'[kp]      (s += chunk[lane q of the row] * w'[kp]; the stream holds -L)
// executed by all lanes alike: 1 LDS load + 1 wait + 16 chain instructions per 16 terms, where a broadcast read per pair of terms
// cost two instructions per term (and 64 registers of prefetch buffers).  DIST chunks are in flight ahead of the chain.
constexpr int kGridBwdDist = 2;
template <int NMAX, int C>
struct GridBwd {
  static constexpr int CH = 16, DIST = kGridBwdDist, TOT = grid_tri1(NMAX), NCH = (TOT + CH - 1) / CH;
  template <class E, class RC>
  static SA_HD __attribute__((always_inline)) void load(E &ex, const double *U, RC (&buf)[DIST + 1]) {
    if constexpr (C < NCH) ex.par([&](int l) { buf[C % (DIST + 1)][l] = U[C * CH + (l & 15)]; });
  }
  // s: running sum of the open row; wr[kp]: finished weights w'[kp]; returns when the last row (n - 1) is done
  template <class E, class RC>
  static SA_HD __attribute__((always_inline)) void run(E &ex, int no, const double *U, double *Wv, RC (&buf)[DIST + 1], double (&wr)[NMAX], double s) {
    if constexpr (C < NCH) {
      GridBwd<NMAX, C + DIST>::load(ex, U, buf);
      const bool done = grid_static_any<0, CH>([&](auto QC) {
        constexpr int q = decltype(QC)::value, t = C * CH + q;          // stream position: row and column are compile-time
        if constexpr (t < TOT) {
          constexpr int ip = grid_row_of(t), pos = t - grid_tri1(ip);
          if constexpr (pos == 0) s = ex.template row_bcast<q>(buf[C % (DIST + 1)]);              // z'[ip] opens the row
          else s = ex.template row_bcast_fma<q>(buf[C % (DIST + 1)], wr[ip - pos], s);           // kp = ip - pos: ip-1 .. 0, i.e. k = i+1 .. n-1 ascending
          if constexpr (pos == ip) {                                    // row complete
            wr[ip] = s;
            Wv[no - 1 - ip] = s;
            if (ip + 1 >= no) return true;
          }
        }
        return false;
      });
      if (done) return;
      GridBwd<NMAX, C + 1>::run(ex, no, U, Wv, buf, wr, s);
    }
  }
};

// slot e (0 .. count-1) of column step K: block column K first (it holds column k + 1, whose values the next pivot needs)
SA_HD constexpr int grid_step_slots(int NB, int K) { return (NB - K) * (NB - K + 1) / 2; }
SA_HD constexpr int grid_step_J(int NB, int K, int e) { int J = K; while (e >= NB - J) { e -= NB - J; J++; } return J; }
SA_HD constexpr int grid_step_I(int NB, int K, int e) { int J = K; while (e >= NB - J) { e -= NB - J; J++; } return J + e; }

// ---- one column step k = 8 K + kk of the factorisation
template <int NB, int K, class E, class RV, class RD>
SA_HD __attribute__((always_inline)) bool grid_factor_step(E &ex, int k, int no, double *U, double *dump, RV &V, RD &sreg, RD &invd_mine) {
  const int kk = k & 7;
  const double dk = ex.lane_bcast_col(V, grid_slot(K, K), 9 * kk);      // pivot D[k] = V[k][k]: lane a == b == kk
  if (dk < 1e-12) return false;
  const double invd = 1.0 / dk;
  const double yk = ex.lane_bcast(sreg, k);                             // forward substitution: y[k] is final (math.h:58-66)
  const int ipk = no - 1 - k;
  const int cb = grid_tri1(ipk) - k;                                    // U[cb + r] = L[r][k], r = k+1 .. no-1
  ex.par([&](int l) {
    const int a = l & 7, b = l >> 3;
    // L[i][k] = lij * invD (math.h:49): the lanes b == kk hold column k; everybody else (and rows outside k+1 .. n-1) stores
    // to a per-lane dump word -- one instruction stream without branches
    const int doff = (int)(dump - U) + l;               // (offsets from ONE base: a select among pointers would go through scratch)
    grid_static_any<K, NB>([&](auto IC) {
      constexpr int I = decltype(IC)::value;
      const int r = 8 * I + a;
      const double lp = V[l].v[grid_slot(I, K)] * invd;
      const int off = (b == kk && r > k && r < no) ? cb + r : doff;
      U[off] = -lp;                                     // the stream holds -L (what the solves multiply with); products of two of them are those of L
      return false;
    });
  });
  ex.wsync();                                       // the column is in LDS (one wave: LDS traffic is in order, only the compiler needs the fence)
  const bool k_even = (k & 1) == 0;
  ex.par([&](int l) {
    const int a = l & 7, b = l >> 3;
    double lr[NB], lc[NB];
    const double *ua = U + cb + a, *ub = U + cb + b;      // one base per operand kind, the block rows at immediate offsets
    grid_static_any<K, NB>([&](auto IC) { constexpr int I = decltype(IC)::value; lr[I] = ua[8 * I]; lc[I] = ub[8 * I]; return false; });
    const double lp = U[cb + l];
    // column j = k + 1 has k + 1 terms; its last one (this one) is fused when that count is odd (k even); it lies in block
    // column K (kk <= 6 when k is even), lanes b == kk + 1
    const bool fz = k_even && b == kk + 1;
    constexpr int NSL = grid_step_slots(NB, K), G = 4;
    grid_static_any<0, (NSL + G - 1) / G>([&](auto GC) {
      constexpr int g0 = decltype(GC)::value * G;
      // four independent chains at a time: products, scaled products, differences (every index a compile-time constant)
      constexpr int I0 = grid_step_I(NB, K, g0), J0 = grid_step_J(NB, K, g0);
      constexpr int I1 = grid_step_I(NB, K, g0 + 1 < NSL ? g0 + 1 : g0), J1 = grid_step_J(NB, K, g0 + 1 < NSL ? g0 + 1 : g0);
      constexpr int I2 = grid_step_I(NB, K, g0 + 2 < NSL ? g0 + 2 : g0), J2 = grid_step_J(NB, K, g0 + 2 < NSL ? g0 + 2 : g0);
      constexpr int I3 = grid_step_I(NB, K, g0 + 3 < NSL ? g0 + 3 : g0), J3 = grid_step_J(NB, K, g0 + 3 < NSL ? g0 + 3 : g0);
      constexpr bool h1 = g0 + 1 < NSL, h2 = g0 + 2 < NSL, h3 = g0 + 3 < NSL;
      double &x0 = V[l].v[grid_slot(I0, J0)], &x1 = V[l].v[grid_slot(I1, J1)], &x2 = V[l].v[grid_slot(I2, J2)], &x3 = V[l].v[grid_slot(I3, J3)];
      const double t0 = lr[I0] * lc[J0], t1 = h1 ? lr[I1] * lc[J1] : 0.0, t2 = h2 ? lr[I2] * lc[J2] : 0.0, t3 = h3 ? lr[I3] * lc[J3] : 0.0;
      double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
      if constexpr (J0 == K) f0 = fma(-t0, dk, x0);
      if constexpr (h1 && J1 == K) f1 = fma(-t1, dk, x1);
      if constexpr (h2 && J2 == K) f2 = fma(-t2, dk, x2);
      if constexpr (h3 && J3 == K) f3 = fma(-t3, dk, x3);
      const double p0 = t0 * dk, p1 = t1 * dk, p2 = t2 * dk, p3 = t3 * dk;
      const double u0 = x0 - p0, u1 = x1 - p1, u2 = x2 - p2, u3 = x3 - p3;
      x0 = (J0 == K && fz) ? f0 : u0;
      if constexpr (h1) x1 = (J1 == K && fz) ? f1 : u1;
      if constexpr (h2) x2 = (J2 == K && fz) ? f2 : u2;
      if constexpr (h3) x3 = (J3 == K && fz) ? f3 : u3;
      SA_PIN_F64(x0);
      if constexpr (h1) SA_PIN_F64(x1);
      if constexpr (h2) SA_PIN_F64(x2);
      if constexpr (h3) SA_PIN_F64(x3);
      return false;
    });
    {   // forward substitution, lane = row (math.h:58-66; the last term of an odd-length chain is fused, canon.h)
      const double v = fold_fused(k, l) ? fma(lp, yk, sreg[l]) : sreg[l] + lp * yk;       // lp = -L[l][k]
      if (l > k && l < no) sreg[l] = v;
      if (l == k) invd_mine[l] = invd;
    }
  });
  return true;
}

template <int NB, int K, class E, class RV, class RD>
SA_HD __attribute__((always_inline)) bool grid_factor_blocks(E &ex, int no, double *U, double *dump, RV &V, RD &sreg, RD &invd_mine) {
  if constexpr (K < NB) {
    int kend = 8 * K + 8 < no ? 8 * K + 8 : no;
    for (int k = 8 * K; k < kend; k++)
      if (!grid_factor_step<NB, K>(ex, k, no, U, dump, V, sreg, invd_mine)) return false;
    if (8 * K + 8 >= no) return true;
    return grid_factor_blocks<NB, K + 1>(ex, no, U, dump, V, sreg, invd_mine);
  } else {
    return true;
  }
}

template <class E, int NB>
SA_HD void ols_stage_grid(E &ex, const ChanParam &p, const int *self, const int *other, int n,
                          double *p_out, char *lds_base, unsigned long long *prof = nullptr) {
  static_assert(E::nl == 64 && NB >= 1 && NB <= 8, "one wave, up to 64 taps");
  constexpr int NMAX = 8 * NB, NS = grid_slot(NB - 1, NB - 1) + 1;
  unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;
#define SA_TICK(i) do { if (prof) { const unsigned long long now_ = E::clock(); tp[i] += now_ - tc; tc = now_; } } while (0)
  const int no = E::uniform(p.n_ols);
  OlsLdsGrid L;
  L.carve(lds_base, NMAX);
  const unsigned long long *exptab = reinterpret_cast<const unsigned long long *>(L.libm + 128 * 3);

  typename E::template Reg<double> xr, breg, sreg, invd_mine, dacc;
  typename E::template Reg<OlsGridRegs<NS>> M, V;
  typename E::template Reg<int> xnext;

  ex.par([&](int l) {
    xr[l] = 0.0; breg[l] = 0.0; sreg[l] = 0.0; invd_mine[l] = 0.0; dacc[l] = 0.0;
#pragma unroll
    for (int q = 0; q < NS; q++) { M[l].v[q] = 0.0; V[l].v[q] = 0.0; }
    if (l < NMAX) L.X[l] = 0.0;
    for (int e = l; e < NMAX + kOlsPad; e += 64) L.Wv[e] = 0.0;
    for (int e = l; e < grid_tri1(NMAX) + 2 * kGridGuard; e += 64) L.U[e - kGridGuard] = 0.0;
    sa_stage_tables(L.libm, l, 64);
    xnext[l] = (l < no && n > 0) ? ols_x(p, self, other, n, 0, l) : 0;
  });
  ex.sync();

  double esum = 0.0;
  int km = 0;
  const double lambda = p.lambda, nu = p.nu_eff;
  const double one_m_lambda = 1.0 - lambda;

  if (prof) tc = E::clock();
  int sv_ahead = n > 0 ? self[0] : 0;              // this step's sample, fetched one step ahead (the load is off the chain)
  for (int t = 0; t < n; t++) {
    const int sv = sv_ahead;
    if (t + 1 < n) sv_ahead = self[t + 1];
    ex.par([&](int l) {
      xr[l] = (double)xnext[l];
      if (l < no) L.X[l] = xr[l];
      if (l < no && t + 1 < n) xnext[l] = ols_x(p, self, other, n, t + 1, l);
    });
    ex.sync();
    double pred = 0.0, val = 0.0, ff = 0.0;
    ex.par([&](int l) {          // slmath::dot with its eight FMA accumulators spread over lanes
      const int a = l & 7;
      double c = 0.0;
      for (int i = 0; i + 8 <= no; i += 8) c = fma(L.X[i + a], L.Wv[i + a], c);
      dacc[l] = c;
    });
    ex.uni([&]() { pred = ols_dot_finish(ex, dacc, L.X, L.Wv, no); });
    ex.par([&](int l) { if (l == 0) p_out[t] = pred; });
    ex.uni([&]() {
      val = (double)sv;
      const double e = val - pred;
      esum = fma(p.beta_sum, esum, fabs(e));
      const double c = sa_pow_t(esum + p.beta_add, -p.beta_pow, L.libm, exptab);
      ff = one_m_lambda * c;
    });
    SA_TICK(0);
    // covariance / rhs update (ols.cpp:38-45): M[i][j] = lambda M[i][j] + ff (x_i x_j) on this lane's slots, b on lane = row
    ex.par([&](int l) {
      const int a = l & 7, b = l >> 3;
      double xa[NB], xb[NB];
#pragma unroll
      for (int I = 0; I < NB; I++) { xa[I] = L.X[8 * I + a]; xb[I] = L.X[8 * I + b]; }
#pragma unroll
      for (int I = 0; I < NB; I++)
#pragma unroll
        for (int J = 0; J <= I; J++) M[l].v[grid_slot(I, J)] = fma(lambda, M[l].v[grid_slot(I, J)], ff * (xa[I] * xb[J]));
      breg[l] = fma(lambda, breg[l], ff * (xr[l] * val));
    });
    SA_TICK(1);
    km++;
    if (km >= p.k) {
      km = 0;
      // working copy of A + nu I (lower triangle; the rest is never read)
      ex.par([&](int l) {
        const int a = l & 7, b = l >> 3;
#pragma unroll
        for (int I = 0; I < NB; I++)
#pragma unroll
          for (int J = 0; J <= I; J++) V[l].v[grid_slot(I, J)] = (I == J && a == b) ? M[l].v[grid_slot(I, J)] + nu : M[l].v[grid_slot(I, J)];
        sreg[l] = breg[l];                       // forward substitution starts from b
      });
      const bool ok = grid_factor_blocks<NB, 0>(ex, no, L.U, L.dump, V, sreg, invd_mine);
      SA_TICK(2);
      if (ok) {
        ex.par([&](int l) { if (l < no) L.U[grid_tri1(no - 1 - l)] = sreg[l] * invd_mine[l]; });     // z[i] = y[i] * invD[i] opens row ip = n-1-i
        SA_TICK(3);
        ex.wsync();
        {
          double wr[NMAX];
          typename E::template Reg<double> buf[kGridBwdDist + 1];
          GridBwd<NMAX, 0>::load(ex, L.U, buf); GridBwd<NMAX, 1>::load(ex, L.U, buf);
          static_assert(kGridBwdDist == 2, "chunks 0..DIST-1 are requested here");
          GridBwd<NMAX, 0>::run(ex, no, L.U, L.Wv, buf, wr, 0.0);
        }
        ex.wsync();
        SA_TICK(4);
      }
    }
    ex.sync();
    SA_TICK(5);
  }
  if (prof) ex.par([&](int l) { if (l == 0) for (int i = 0; i < 8; i++) prof[i] = tp[i]; });
#undef SA_TICK
}

}  // namespace sacamd
