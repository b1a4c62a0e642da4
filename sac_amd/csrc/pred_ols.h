// sac_amd/csrc/pred_ols.h -- stage 1 of the predictor: the OLS recurrence.
//
// Reference: OLS (/root/reference/src/pred/ols.cpp:7-57) + slmath::LDLT
// (common/math.h:14-78) + RunSumGEO (common/utils.h:39-72), driven by
// Predictor::fillbuf_ch0/ch1 (libsac/pred.cpp:17-31).
//
// Encoder-side observation this design rests on: the OLS stage depends only on the input
// PCM (its regressor and its target are samples, never predictions), so for every
// (frame x candidate x channel) it is a self-contained recurrence that can run ahead of the
// cascade and bias stages and hand p_lpc[t] over through HBM.
//
// One workgroup per work-item: one wave up to 32 taps (ols_stage_fast), four waves for 33..64
// (ols_stage_panel) and 65..96 (ols_stage_panel2, two matrix rows per lane).  Lane = matrix row;
// the covariance estimate (packed lower triangle) and L (column-major, zero padded) live in LDS.
// Whatever the schedule, every element sees exactly the reference's left-looking subtraction
// chain -- same terms, same order, same fused/unfused pattern (canon.h) -- so p_lpc is
// bit-identical to the reference build.
#pragma once
#include "canon.h"
#include "libm_port.h"
#include "params.h"

namespace sacamd {

SA_HD int tri_off(int n, int j) { return j * n - (j * (j - 1)) / 2; }   // start of column j
SA_HD int tri_count(int n) { return n * (n + 1) / 2; }

// regressor element j at step t (zero outside the window), pred.cpp:17-31
SA_HD int ols_x(const ChanParam &p, const int *self, const int *other, int n, int t, int j) {
  if (j < p.a) {
    const int i = t - p.a + j;
    return i >= 0 ? self[i] : 0;
  }
  int u = t - p.du;
  if (u < 0) u = 0;
  const int i = u - p.b + (j - p.a);
  return (i >= 0 && i < n) ? other[i] : 0;
}

// ---------------------------------------------------------------------------------------------
// One-wave workgroups (E::nl == 64, n_ols <= 64; launched for <= 32 taps):
//  * x, b and the forward-solve vector: one element per lane in registers, moved with v_readlane
//    broadcasts instead of LDS round trips + barriers;
//  * LDL^T left-looking by columns, chains in whole 8-term chunks (see ols_stage_fast);
//  * back-substitution fully unrolled, anchored at the last row (OlsBwdRows).
struct DArr4 { double v[4]; };
constexpr int kOlsPad = 8;

// rows ip = IP .. NMAX-1 of the end-anchored back-substitution (template recursion: the register
// array wr is only ever indexed by compile-time constants)
// (build-time knob, tools/build_variant.sh; 16 in rounds 1-3.  8 frees 14-32 VGPRs in the one-wave kernels at the same throughput)
#ifndef SACAMD_EXP_BWD_CHUNK
#define SACAMD_EXP_BWD_CHUNK 8
#endif
constexpr int kOlsBwdChunk = SACAMD_EXP_BWD_CHUNK;
template <int NMAX, int S, int IP>
struct OlsBwdRows {
  static SA_HD __attribute__((always_inline)) void run(int no, const double *lds0, const double *lb, const double *zb, double *wb, double (&wr)[NMAX]) {
    if constexpr (IP < NMAX) {
      if (IP >= no) return;
      double s_ = zb[NMAX - 1 - IP];
      // one opaque base per row: the row's elements then sit within the 8-bit offset range of
      // ds_read2_b64 and the compiler does not materialise one address per pair
      int rowo = (int)(lb - lds0) + (NMAX - 1 - IP) * S;   // whole offset from the start of LDS: nothing left to fold into immediates
      SA_OPAQUE_INT(rowo);
      const double *rowp = lds0 + rowo;
      // the row's L elements are requested kOlsBwdChunk at a time, one chunk ahead of the chain
      constexpr int CH = kOlsBwdChunk;
      constexpr int NCH = (IP + CH - 1) / CH;
      double buf[2][CH];
#pragma unroll
      for (int q = 0; q < CH; q++) if (q < IP) buf[0][q] = rowp[NMAX - 1 - (IP - 1 - q)];
#pragma unroll
      for (int c = 0; c < NCH; c++) {
#pragma unroll
        for (int q = 0; q < CH; q++) if ((c + 1) * CH + q < IP) buf[(c + 1) & 1][q] = rowp[NMAX - 1 - (IP - 1 - ((c + 1) * CH + q))];
#pragma unroll
        for (int q = 0; q < CH; q++) if (c * CH + q < IP) s_ = fma(-buf[c & 1][q], wr[IP - 1 - (c * CH + q)], s_);
      }
      wr[IP] = s_;
      wb[NMAX - 1 - IP] = s_;
      OlsBwdRows<NMAX, S, IP + 1>::run(no, lds0, lb, zb, wb, wr);
    }
  }
};

// second half of slmath::dot (common/math.h:130-161) once lanes 0..7 hold its eight accumulators:
// s_c + t_c, ((s0+s1)+s2)+s3, then the < 8 tail (transform_reduce order)
template <class E, class R>
SA_HD double ols_dot_finish(E &ex, const R &dacc, const double *x, const double *y, int n) {
  double total = 0.0;
  const int nb = n & ~7;
  if (nb) {
    const double s0 = ex.lane_bcast(dacc, 0) + ex.lane_bcast(dacc, 4), s1 = ex.lane_bcast(dacc, 1) + ex.lane_bcast(dacc, 5);
    const double s2 = ex.lane_bcast(dacc, 2) + ex.lane_bcast(dacc, 6), s3 = ex.lane_bcast(dacc, 3) + ex.lane_bcast(dacc, 7);
    total = ((s0 + s1) + s2) + s3;
  }
  total += tr_dot(x + nb, y + nb, n - nb);
  return total;
}

struct OlsLdsFast {
  double *X, *Wv, *Dv, *M, *Lq, *libm, *dump;
  // Lq: L stored column by column, [column k][row i] with row stride SP = NMAX + kOlsPad.  Rows
  // >= n_ols of every column (including the kOlsPad padding rows) and rows <= k of column k are
  // never written and stay 0.0: the solve loops run whole 8-term chunks and rely on those zeros
  // (a term with a zero factor leaves the fused chain unchanged), so they carry no masks.  Wv is
  // padded the same way.  M stays a packed triangle.
  SA_HD static size_t bytes(int nmax) {
    return (size_t)(nmax + 2 * (nmax + kOlsPad) + tri_count(nmax) + nmax * (nmax + kOlsPad) + kLibmLdsDoubles + 2) * sizeof(double) + 16;
  }
  SA_HD void carve(char *base, int nmax) {
    double *d = reinterpret_cast<double *>(base);
    // Lq comes last and Dv / Wv come after >= 16 doubles: the back-substitution addresses rows and
    // columns relative to the LAST row (n_ols - 1) with compile-time offsets, so its base pointers
    // sit up to (NMAX - n_ols) * (stride + 1) elements below the arrays themselves
    X = d; d += nmax; Wv = d; d += nmax + kOlsPad; Dv = d; d += nmax + kOlsPad;
    M = d; d += tri_count(nmax); libm = d; d += kLibmLdsDoubles; dump = d; d += 2;
    Lq = d; d += nmax * (nmax + kOlsPad);
  }
};

// decoder: number of decoded samples of the OTHER channel the regressor of step t reads (0: none)
SA_HD int ols_other_need(const ChanParam &p, int n, int t) {
  if (p.b + p.c <= 0) return 0;
  int u = t - p.du;
  if (u < 0) u = 0;
  const int need = u + p.c;                 // highest index read is u - b + (b + c - 1)
  return need < n ? need : n;
}

// ---------------------------------------------------------------------------------------------
// Register-resident one-wave kernel (round 3; launched for <= 64 taps).  Lane l <-> row l.  The working row V[l][*] of the
// factorisation lives in REGISTERS and the LDL^T runs RIGHT-looking: as soon as column k is final, its term is applied to
// every later column j,
//     V[i][j] -= (L[i][k] * L[j][k]) * D[k]        (fused for the last term of an odd column, canon.h fold_add)
// with L[j][k] broadcast from lane j (v_readlane: a scalar operand) -- no LDS traffic on the chain at all, where
// ols_stage_fast pays three LDS reads per term.  Every element still sees its terms in ascending k with the reference's
// fused / unfused pattern, so p_lpc stays bit-identical.
// The column loop is a RUN-TIME loop over k whose body is unrolled over register slots: slot q holds column k + q, and the
// update writes column k + 1 + q's new value into slot q (V[q] = V[q+1] - term), so the pivot column is always slot 0,
// register indices are compile-time constants, and the code is a few KB whatever the regressor length.
// The covariance row M[l][*] is register-resident too (MREG: since round 4 for every class; the packed-triangle-in-LDS path of
// rounds 1-3 remains selectable at build time, SACAMD_EXP_MREG_MAX).  The forward substitution rides along as before; the backward substitution is the unrolled end-anchored
// chain over L in LDS (OlsBwdRows).
template <int N> struct OlsRow { double v[N]; };

// slots Q .. NMAX-2 of one column step: V[q] = V[q+1] - (lk * L[k+1+q][k]) * dk, in aligned groups of four slots under one
// test of the group's first column against the regressor length (slots beyond it only hold values nobody reads; the four
// independent chains of a group interleave)
template <int NMAX, int Q>
struct OlsRankOne {
  // bj[u] = L[k+1+Q+u][k] of this group, requested by the previous group (or the caller): the LDS round trip of a group's
  // broadcast reads runs under the arithmetic of the group before it
  template <class E, class RV, class RD>
  static SA_HD __attribute__((always_inline)) void run(E &ex, int rem /* columns after k */, const double *lrow /* &L[k+1][k] */, bool k_even, RV &V, const RD &lk, double dk,
                                                       const double (&bj)[4]) {
    if constexpr (Q < NMAX - 1) {
      if (Q < rem) {
        constexpr int QE = (Q + 4) < (NMAX - 1) ? (Q + 4) : (NMAX - 1);
        // L[j][k], j = k + 1 + slot: read back from the column just stored to LDS -- every lane the same address (a broadcast
        // read, two columns per ds_read2_b64) instead of two v_readlane per column.  Reads beyond the regressor length stay
        // inside the column's zero padding.
        double bn[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (QE < NMAX - 1) {
#pragma unroll
          for (int u = 0; u < 4; u++) bn[u] = lrow[QE + u];
        }
        ex.par([&](int l) {
          double tt[4];
#pragma unroll
          for (int u = 0; u < 4; u++) if (Q + u < QE) tt[u] = lk[l] * bj[u];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (Q + u < QE) {
              // column j = k + 1 has k + 1 terms; its last one (this one) is fused when that count is odd (canon.h fold_add)
              if (Q + u == 0) V[l].v[0] = k_even ? fma(-tt[0], dk, V[l].v[1]) : V[l].v[1] - tt[0] * dk;
              else V[l].v[Q + u] = V[l].v[Q + u + 1] - tt[u] * dk;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) if (Q + u < QE) SA_PIN_F64(V[l].v[Q + u]);
        });
        OlsRankOne<NMAX, QE>::run(ex, rem, lrow, k_even, V, lk, dk, bn);
      }
    }
  }
};

// covariance update of columns J .. NMAX-1 (ols.cpp:38-42): M[i][j] = lambda M[i][j] + ff (x_i x_j), x_j broadcast from lane j;
// groups of four columns as in OlsRankOne
template <int NMAX, int J>
struct OlsCovUpdate {
  template <class E, class RV, class RD>
  static SA_HD __attribute__((always_inline)) void run(E &ex, int no, RV &M, const RD &xr, double lambda, double ff) {
    if constexpr (J < NMAX) {
      // a fresh copy of the (uniform) length now and then: otherwise the compiler evaluates all "j < no" tests of the unrolled
      // code once, ahead of the sample loop, and keeps their lane masks alive (spilled to VGPR lanes)
      if constexpr ((J & 7) == 0) SA_OPAQUE_SINT(no);
      if (J < no) {
        constexpr int JE = (J + 4) < NMAX ? (J + 4) : NMAX;
        double xj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) if (J + u < JE) xj[u] = ex.lane_bcast(xr, J + u);
        ex.par([&](int l) {
#pragma unroll
          for (int u = 0; u < 4; u++) if (J + u < JE) M[l].v[J + u] = fma(lambda, M[l].v[J + u], ff * (xr[l] * xj[u]));
#pragma unroll
          for (int u = 0; u < 4; u++) if (J + u < JE) SA_PIN_F64(M[l].v[J + u]);
        });
        OlsCovUpdate<NMAX, JE>::run(ex, no, M, xr, lambda, ff);
      }
    }
  }
};

template <class E, int NMAX>
SA_HD void ols_stage_reg(E &ex, const ChanParam &p, const int *self, const int *other, int n,
                         double *p_out, char *lds_base, unsigned long long *prof = nullptr, const DecLink *dec = nullptr) {
  static_assert(E::nl == 64 && NMAX <= 64 && NMAX % 4 == 0, "one-wave path");
// Round 4: the covariance rows are register resident for EVERY one-wave class (rounds 1-3: up to 32 taps, a packed triangle in LDS
// above).  The LDS path's address arithmetic cost more registers than the rows themselves (64 taps: 322 instead of 330 registers) and
// a fifth of the time: 116 -> 143, 86 -> 106, 50 -> 62, 29 -> 35 M item-steps/s at 40 / 48 / 56 / 64 taps (profiles/r04/throughput_mreg_*.txt).
#ifndef SACAMD_EXP_MREG_MAX
#define SACAMD_EXP_MREG_MAX 64
#endif
  constexpr bool MREG = NMAX <= SACAMD_EXP_MREG_MAX;         // covariance rows in registers (else: packed triangle in LDS)
  constexpr int S = NMAX + kOlsPad;
  unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;
#define SA_TICK(i) do { if (prof) { const unsigned long long now_ = E::clock(); tp[i] += now_ - tc; tc = now_; } } while (0)
  constexpr int NL = 64;
  const int no = E::uniform(p.n_ols);
  const int ntri = tri_count(no);
  OlsLdsFast L;
  L.carve(lds_base, NMAX);
  const unsigned long long *exptab = reinterpret_cast<const unsigned long long *>(L.libm + 128 * 3);

  typename E::template Reg<double> xr, breg, sreg, zreg, invd_mine, lk, dacc;
  typename E::template Reg<OlsRow<MREG ? NMAX : 1>> M;
  typename E::template Reg<OlsRow<NMAX>> V;
  typename E::template Reg<int> xnext;

  if (dec) sa_wait_ge(dec->prog_other, ols_other_need(p, n, 0), dec->fail);     // decoder: the first regressor may read the partner's first samples
  ex.par([&](int l) {
    xr[l] = 0.0; breg[l] = 0.0; sreg[l] = 0.0; zreg[l] = 0.0; invd_mine[l] = 0.0; lk[l] = 0.0; dacc[l] = 0.0;
#pragma unroll
    for (int j = 0; j < NMAX; j++) { if (MREG) M[l].v[j] = 0.0; V[l].v[j] = 0.0; }
    if (l < no) L.X[l] = 0.0;
    for (int e = l; e < NMAX + kOlsPad; e += NL) { L.Wv[e] = 0.0; L.Dv[e] = 0.0; }
    if (!MREG) for (int e = l; e < ntri; e += NL) L.M[e] = 0.0;
    for (int e = l; e < NMAX * S; e += NL) L.Lq[e] = 0.0;
    sa_stage_tables(L.libm, l, NL);
    xnext[l] = (l < no && n > 0) ? ols_x(p, self, other, n, 0, l) : 0;
  });
  ex.sync();

  double esum = 0.0;
  int km = 0;
  const double lambda = p.lambda, nu = p.nu_eff;
  const double one_m_lambda = 1.0 - lambda;

  if (prof) tc = E::clock();
  int sv_ahead = (n > 0 && !dec) ? self[0] : 0;              // this step's sample, fetched one step ahead (the load is off the chain)
  for (int t = 0; t < n; t++) {
    int sv = sv_ahead;
    if (!dec && t + 1 < n) sv_ahead = self[t + 1];
    ex.par([&](int l) {
      xr[l] = (double)xnext[l];
      if (l < no) L.X[l] = xr[l];
      if (!dec && l < no && t + 1 < n) xnext[l] = ols_x(p, self, other, n, t + 1, l);
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
    ex.par([&](int l) { if (l == 0) { p_out[t] = pred; if (dec) sa_publish(dec->prog_out, t + 1); } });
    if (dec) {         // decoder: the sample exists once the cascade and bias stages have added their part to this prediction
      if (!sa_wait_ge(dec->prog_self, t + 1, dec->fail)) return;
      sv = self[t];
    }
    ex.uni([&]() {
      val = (double)sv;
      const double e = val - pred;
      esum = fma(p.beta_sum, esum, fabs(e));
      const double c = sa_pow_t(esum + p.beta_add, -p.beta_pow, L.libm, exptab);
      ff = one_m_lambda * c;
    });
    SA_TICK(0);
    if constexpr (MREG) {
      OlsCovUpdate<NMAX, 0>::run(ex, no, M, xr, lambda, ff);
      ex.par([&](int l) { breg[l] = fma(lambda, breg[l], ff * (xr[l] * val)); });
    } else {
      // covariance / rhs update on the packed triangle in LDS (ols.cpp:38-45): lane = row i, loop over columns j <= i.
      // Loads are unconditional (rows above the diagonal read harmless neighbours), only the stores are masked.
      ex.par([&](int l) {
        if (l < no) {
          const double xi = xr[l];
          int j = 0;
          double *dump = L.dump;                       // write-only slot
          for (; j + 8 <= no; j += 8) {
            double m[8], xj[8];
            int e[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { e[u] = tri_off(no, j + u) + (l - (j + u)); xj[u] = L.X[j + u]; m[u] = L.M[e[u]]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const double v = fma(lambda, m[u], ff * (xi * xj[u]));
              double *dst = (l >= j + u) ? &L.M[e[u]] : dump;      // select the address, not the lane
              *dst = v;
            }
          }
          for (; j < no; j++) { const int e = tri_off(no, j) + (l - j); const double v = fma(lambda, L.M[e], ff * (xi * L.X[j])); if (l >= j) L.M[e] = v; }
          breg[l] = fma(lambda, breg[l], ff * (xi * val));
        }
      });
    }
    SA_TICK(1);
    km++;
    if (km >= p.k) {
      km = 0;
      // working copy: slot j = column j of A + nu I (lower triangle; the rest is never read)
      if constexpr (MREG) {
        ex.par([&](int l) {
#pragma unroll
          for (int j = 0; j < NMAX; j++) V[l].v[j] = (l == j) ? M[l].v[j] + nu : M[l].v[j];
        });
      } else {
        ex.wsync();
        ex.par([&](int l) {
          const int lm = l < no ? l : no - 1;
          int oj = 0;                                 // tri_off(no, j)
#pragma unroll
          for (int j = 0; j < NMAX; j++) {
            if (j < no) {
              const double m = L.M[oj + (lm - j)];
              V[l].v[j] = (l == j) ? m + nu : m;
              oj += no - j;
            }
          }
        });
      }
      ex.par([&](int l) { sreg[l] = breg[l]; });     // forward substitution starts from b
      bool ok = true;
      for (int k = 0; k < no; k++) {
        const double dk = ex.lane_bcast_col(V, 0, k);              // pivot D[k] = V[k][k] (slot 0 = column k)
        if (dk < 1e-12) { ok = false; break; }
        const double invd = 1.0 / dk;
        const double yk = ex.lane_bcast(sreg, k);                  // forward substitution: y[k] is final (math.h:58-66)
        ex.par([&](int l) {
          const double lp = V[l].v[0] * invd;                      // L[i][k] = lij * invD (math.h:49)
          lk[l] = lp;
          if (l > k && l < no) L.Lq[k * S + l] = lp;
          if (l == k) invd_mine[l] = invd;
          const double v = fold_fused(k, l) ? fma(-lp, yk, sreg[l]) : sreg[l] - lp * yk;
          if (l > k && l < no) sreg[l] = v;
        });
        ex.wsync();                                   // the column is in LDS (one wave: LDS traffic is in order, only the compiler needs the fence)
        int rem = no - 1 - k;
        SA_OPAQUE_SINT(rem);
        const double *lrow = L.Lq + k * S + k + 1;
        const double b0[4] = {lrow[0], lrow[1], lrow[2], lrow[3]};
        OlsRankOne<NMAX, 0>::run(ex, rem, lrow, (k & 1) == 0, V, lk, dk, b0);
      }
      SA_TICK(2);
      if (ok) {
        ex.par([&](int l) { zreg[l] = sreg[l] * invd_mine[l]; });
        SA_TICK(3);
        ex.par([&](int l) { if (l < no) L.Dv[l] = zreg[l]; });     // z into LDS for the chain below
        ex.wsync();
        ex.lane0([&]() {
          double wr[NMAX];
          const double *lb = L.Lq - (NMAX - no) * (S + 1);          // lb[(NMAX-1-ip)*S + (NMAX-1-kp)] == L[k][i]
          const double *zb = L.Dv - (NMAX - no);                    // zb[NMAX-1-ip] == z[i]
          double *wb = L.Wv - (NMAX - no);
          OlsBwdRows<NMAX, S, 0>::run(no, L.X, lb, zb, wb, wr);
        });
        ex.wsync();
        SA_TICK(4);
      }
    }
    if (dec && t + 1 < n) {      // decoder: the next regressor, now that this channel's sample t (and the partner's share) exists
      if (!sa_wait_ge(dec->prog_other, ols_other_need(p, n, t + 1), dec->fail)) return;
      ex.par([&](int l) { if (l < no) xnext[l] = ols_x(p, self, other, n, t + 1, l); });
    }
    ex.sync();
    SA_TICK(5);
  }
  if (prof) ex.par([&](int l) { if (l == 0) for (int i = 0; i < 8; i++) prof[i] = tp[i]; });
#undef SA_TICK
}

// ---------------------------------------------------------------------------------------------
// Multi-wave variant of ols_stage_fast for the long regressors (E::nl == 64*PW with PW = 4 or 8,
// n_ols <= 64): same arithmetic, element for element, but the LDL^T runs as a blocked left-looking
// factorisation with panels of PW columns:
//   phase 1  wave w takes column p+w of the panel starting at p (a multiple of PW) and runs its
//            chain over the finished columns k < p in whole 8-term chunks (for PW = 4 the last
//            chunk may reach into the panel: those D[k] are still 0, see ols_stage_fast);
//   phase 2  wave 0 finishes the panel right-looking: column by column it takes the pivot,
//            divides, publishes the scaled column, and immediately applies that column's term to
//            the later columns of the panel -- so every column still sees its terms in ascending
//            k, and the only possibly fused term (k = j-1 with j odd) is always an in-panel one.
// The covariance update is split over the waves by column groups; regressor, prediction and the
// two triangular solves stay on wave 0.
template <int N> struct OlsArr { double v[N]; };

template <class E, int NMAX>
SA_HD void ols_stage_panel(E &ex, const ChanParam &p, const int *self, const int *other, int n,
                           double *p_out, char *lds_base, unsigned long long *prof = nullptr) {
  constexpr int NL = E::nl;
  constexpr int PW = NL / 64;
  static_assert(PW == 4 || PW == 8, "panel width = number of waves");
  constexpr int S = NMAX + kOlsPad;
  unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;
#define SA_TICK(i) do { if (prof) { const unsigned long long now_ = E::clock(); tp[i] += now_ - tc; tc = now_; } } while (0)
  const int no = p.n_ols;
  const int ntri = tri_count(no);
  OlsLdsFast L;
  L.carve(lds_base, NMAX);
  double *ACC = L.Lq + NMAX * S;     // [PW][64] phase-1 results
  double *sc = ACC + PW * 64;        // [0] forgetting factor of the step, [1] factorisation ok flag
  const unsigned long long *exptab = reinterpret_cast<const unsigned long long *>(L.libm + 128 * 3);

  typename E::template Reg<double> xr, breg, sreg, zreg, invd_mine, accc, lpc, dacc;
  typename E::template Reg<OlsArr<PW>> accp;
  typename E::template Reg<int> xnext;

  ex.par([&](int l) {
    xr[l] = 0.0; breg[l] = 0.0; sreg[l] = 0.0; zreg[l] = 0.0; invd_mine[l] = 0.0; accc[l] = 0.0; lpc[l] = 0.0; dacc[l] = 0.0;
    for (int c = 0; c < PW; c++) accp[l].v[c] = 0.0;
    if (l < no) L.X[l] = 0.0;
    for (int e = l; e < NMAX + kOlsPad; e += NL) { L.Wv[e] = 0.0; L.Dv[e] = 0.0; }
    for (int e = l; e < ntri; e += NL) L.M[e] = 0.0;
    for (int e = l; e < NMAX * S; e += NL) L.Lq[e] = 0.0;
    ACC[l] = 0.0;
    if (l < 4) sc[l] = 0.0;
    sa_stage_tables(L.libm, l, NL);
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
      if (l < 64) {
        xr[l] = (double)xnext[l];
        if (l < no) L.X[l] = xr[l];
        if (l < no && t + 1 < n) xnext[l] = ols_x(p, self, other, n, t + 1, l);
      }
    });
    ex.sync();
    double pred = 0.0, val = 0.0, ff = 0.0;
    ex.leader_par([&](int l) {
      const int a = l & 7;
      double c = 0.0;
      for (int i = 0; i + 8 <= no; i += 8) c = fma(L.X[i + a], L.Wv[i + a], c);
      dacc[l] = c;
    });
    ex.leader([&]() {
      pred = ols_dot_finish(ex, dacc, L.X, L.Wv, no);
      val = (double)sv;
      const double e = val - pred;
      esum = fma(p.beta_sum, esum, fabs(e));
      const double c = sa_pow_t(esum + p.beta_add, -p.beta_pow, L.libm, exptab);
      ff = one_m_lambda * c;
    });
    ex.par([&](int l) { if (l == 0) { p_out[t] = pred; sc[0] = ff; } });
    ex.sync();
    SA_TICK(0);
    // covariance / rhs update (ols.cpp:38-45): lane = row, wave w takes the column groups 8w, 8w+8*PW, ..
    ex.par([&](int l) {
      const int w = l >> 6, r = l & 63;
      if (r < no) {
        const double ffl = sc[0];
        const double xi = L.X[r];
        double *dump = L.dump;
        for (int j = 8 * w; j < no; j += 8 * PW) {
          double m[8], xj[8];
          int e[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int jj = j + u < no ? j + u : no - 1;
            e[u] = tri_off(no, jj) + (r - jj); xj[u] = L.X[jj]; m[u] = L.M[e[u]];
          }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const double v = fma(lambda, m[u], ffl * (xi * xj[u]));
            double *dst = (r >= j + u && j + u < no) ? &L.M[e[u]] : dump;
            *dst = v;
          }
        }
        if (w == 0) breg[l] = fma(lambda, breg[l], ff * (xi * val));
      }
    });
    SA_TICK(1);
    km++;
    if (km >= p.k) {
      km = 0;
      ex.par([&](int l) {
        for (int e = l; e < NMAX + kOlsPad; e += NL) L.Dv[e] = 0.0;
        if (l == 0) sc[1] = 1.0;
      });
      ex.sync();                                   // M complete, D cleared
      bool ok = true;
      ex.leader_par([&](int l) { sreg[l] = breg[l]; });   // forward substitution starts from b
      for (int p4 = 0; p4 < no; p4 += PW) {
        const int nchunk = (p4 + 7) >> 3;           // chunks cover k < 8*nchunk <= NMAX; k >= p4 is masked by D == 0
        ex.par([&](int l) {
          const int w = l >> 6, r = l & 63;
          const int j = p4 + w;
          if (j < no) {
            const int rm = r < no ? r : no - 1;
            double s_ = L.M[tri_off(no, j) + (rm - j)];
            if (r == j) s_ = s_ + nu;
            const double *pa = L.Lq + r;              // own row: element k at pa[k*S]
            const double *pb = L.Lq + j;              // row j
            struct Fc { double a[8], b[8], d[8]; };
            auto ld = [&](Fc &c, int m) {
              const int k = 8 * m;
#pragma unroll
              for (int u = 0; u < 8; u++) { c.a[u] = pa[(k + u) * S]; c.b[u] = pb[(k + u) * S]; c.d[u] = L.Dv[k + u]; }
            };
            auto ac = [&](double v, const Fc &c) {
              double td[8];
#pragma unroll
              for (int u = 0; u < 8; u++) td[u] = c.a[u] * c.b[u];
#pragma unroll
              for (int u = 0; u < 8; u++) td[u] = td[u] * c.d[u];
#pragma unroll
              for (int u = 0; u < 8; u++) v = v - td[u];
              return v;
            };
            if (nchunk > 0) {
              Fc A, B;
              ld(A, 0);
              int m = 0;
              while (true) {
                if (m + 1 < nchunk) ld(B, m + 1);
                s_ = ac(s_, A);
                if (++m >= nchunk) break;
                if (m + 1 < nchunk) ld(A, m + 1);
                s_ = ac(s_, B);
                if (++m >= nchunk) break;
              }
            }
            ACC[l] = s_;
          }
        });
        ex.sync();
        if (ex.is_leader()) {
          // phase 2 (wave 0): right-looking inside the panel.  The other waves wait at the barrier
          // below, so the pivots can be published as they are produced.
          ex.leader_par([&](int l) {
#pragma unroll
            for (int c = 0; c < PW; c++) accp[l].v[c] = ACC[c * 64 + l];
          });
#pragma unroll
          for (int c = 0; c < PW; c++) {
            const int j = p4 + c;
            if (j >= no || !ok) break;
            ex.leader_par([&](int l) { accc[l] = accp[l].v[c]; });
            const double dj = ex.lane_bcast(accc, j);
            if (dj < 1e-12) { ok = false; break; }
            const double invd = 1.0 / dj;
            const double yk = ex.lane_bcast(sreg, j);      // forward substitution rides along (see ols_stage_fast)
            ex.leader_par([&](int l) {
              const double v = accc[l] * invd;
              lpc[l] = v;
              if (l > j && l < no) L.Lq[j * S + l] = v;
              if (l == j) invd_mine[l] = invd;
              if (l == 0) L.Dv[j] = dj;
              const double f = fold_fused(j, l) ? fma(-v, yk, sreg[l]) : sreg[l] - v * yk;
              if (l > j && l < no) sreg[l] = f;
            });
            // term k = j of the later columns of this panel (their next term in ascending k)
#pragma unroll
            for (int c2 = c + 1; c2 < PW; c2++) {
              const int j2 = p4 + c2;
              const int jb = j2 < 64 ? j2 : 63;
              const double bq = ex.lane_bcast(lpc, jb);
              const bool fz = (c2 == c + 1) && (j2 & 1);          // k = j2-1 with j2 odd: the fused last term
              ex.leader_par([&](int l) {
                const double tt = lpc[l] * bq;
                accp[l].v[c2] = fz ? fma(-tt, dj, accp[l].v[c2]) : accp[l].v[c2] - tt * dj;
              });
            }
          }
          ex.leader_par([&](int l) { if (l == 0 && !ok) sc[1] = 0.0; });
        }
        ex.sync();
        ok = sc[1] != 0.0;
        if (!ok) break;
      }
      SA_TICK(2);
      if (ok && ex.is_leader()) {
        // (the forward substitution was carried along by the factorisation)
        ex.leader_par([&](int l) { zreg[l] = sreg[l] * invd_mine[l]; });
        SA_TICK(3);
        ex.leader_par([&](int l) { if (l < no) L.Dv[l] = zreg[l]; });     // z into LDS (D is no longer needed)
        ex.wsync();
        ex.lane0([&]() {
          double wr[NMAX];
          const double *lb = L.Lq - (NMAX - no) * (S + 1);
          const double *zb = L.Dv - (NMAX - no);
          double *wb = L.Wv - (NMAX - no);
          OlsBwdRows<NMAX, S, 0>::run(no, L.X, lb, zb, wb, wr);
        });
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

// ---------------------------------------------------------------------------------------------
// Regressors of 65..128 taps: the panel factorisation of ols_stage_panel with TWO matrix rows per
// lane (rows r and r + 64 of lane r).  Same element-by-element arithmetic; every per-row register
// becomes a pair, broadcasts pick the half that owns the row.  One workgroup (four waves) per CU.
template <class E, int NMAX>
SA_HD void ols_stage_panel2(E &ex, const ChanParam &p, const int *self, const int *other, int n,
                            double *p_out, char *lds_base, unsigned long long *prof = nullptr, const DecLink *dec = nullptr) {
  constexpr int NL = E::nl;
  constexpr int PW = NL / 64;
  static_assert(PW == 4 && NMAX > 64 && NMAX <= 128, "four waves, two rows per lane");
  constexpr int S = NMAX + kOlsPad;
  unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;
#define SA_TICK(i) do { if (prof) { const unsigned long long now_ = E::clock(); tp[i] += now_ - tc; tc = now_; } } while (0)
  const int no = p.n_ols;
  const int ntri = tri_count(no);
  OlsLdsFast L;
  L.carve(lds_base, NMAX);
  double *ACC = L.Lq + NMAX * S;     // [PW][2][64] phase-1 results
  double *sc = ACC + PW * 2 * 64;    // [0] forgetting factor of the step, [1] factorisation ok flag
  const unsigned long long *exptab = reinterpret_cast<const unsigned long long *>(L.libm + 128 * 3);

  typedef OlsArr<2> D2;
  typename E::template Reg<D2> xr, breg, sreg, zreg, invd_mine, accc, lpc;
  typename E::template Reg<OlsArr<2 * PW>> accp;
  typename E::template Reg<double> dacc, tmpb;
  struct I2 { int v[2]; };
  typename E::template Reg<I2> xnext;

  if (dec) sa_wait_ge(dec->prog_other, ols_other_need(p, n, 0), dec->fail);     // decoder: the first regressor may read the partner's first samples
  ex.par([&](int l) {
    for (int h = 0; h < 2; h++) {
      xr[l].v[h] = 0.0; breg[l].v[h] = 0.0; sreg[l].v[h] = 0.0; zreg[l].v[h] = 0.0; invd_mine[l].v[h] = 0.0; accc[l].v[h] = 0.0; lpc[l].v[h] = 0.0;
      const int row = l + 64 * h;
      xnext[l].v[h] = (l < 64 && row < no && n > 0) ? ols_x(p, self, other, n, 0, row) : 0;
    }
    for (int c = 0; c < 2 * PW; c++) accp[l].v[c] = 0.0;
    dacc[l] = 0.0; tmpb[l] = 0.0;
    for (int e = l; e < NMAX; e += NL) L.X[e] = 0.0;
    for (int e = l; e < NMAX + kOlsPad; e += NL) { L.Wv[e] = 0.0; L.Dv[e] = 0.0; }
    for (int e = l; e < ntri; e += NL) L.M[e] = 0.0;
    for (int e = l; e < NMAX * S; e += NL) L.Lq[e] = 0.0;
    for (int e = l; e < PW * 2 * 64; e += NL) ACC[e] = 0.0;
    if (l < 4) sc[l] = 0.0;
    sa_stage_tables(L.libm, l, NL);
  });
  ex.sync();

  double esum = 0.0;
  int km = 0;
  const double lambda = p.lambda, nu = p.nu_eff;
  const double one_m_lambda = 1.0 - lambda;

  if (prof) tc = E::clock();
  int sv_ahead = (n > 0 && !dec) ? self[0] : 0;              // this step's sample, fetched one step ahead (the load is off the chain)
  for (int t = 0; t < n; t++) {
    int sv = sv_ahead;
    if (!dec && t + 1 < n) sv_ahead = self[t + 1];
    ex.par([&](int l) {
      if (l < 64) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int row = l + 64 * h;
          xr[l].v[h] = (double)xnext[l].v[h];
          if (row < no) L.X[row] = xr[l].v[h];
          if (!dec && row < no && t + 1 < n) xnext[l].v[h] = ols_x(p, self, other, n, t + 1, row);
        }
      }
    });
    ex.sync();
    double pred = 0.0, val = 0.0, ff = 0.0;
    ex.leader_par([&](int l) {
      const int a = l & 7;
      double c = 0.0;
      for (int i = 0; i + 8 <= no; i += 8) c = fma(L.X[i + a], L.Wv[i + a], c);
      dacc[l] = c;
    });
    ex.leader([&]() { pred = ols_dot_finish(ex, dacc, L.X, L.Wv, no); });
    ex.par([&](int l) { if (l == 0) { p_out[t] = pred; if (dec) sa_publish(dec->prog_out, t + 1); } });
    if (dec) {         // decoder: see ols_stage_reg
      if (!sa_wait_ge(dec->prog_self, t + 1, dec->fail)) return;
      sv = self[t];
    }
    ex.leader([&]() {
      val = (double)sv;
      const double e = val - pred;
      esum = fma(p.beta_sum, esum, fabs(e));
      const double c = sa_pow_t(esum + p.beta_add, -p.beta_pow, L.libm, exptab);
      ff = one_m_lambda * c;
    });
    ex.par([&](int l) { if (l == 0) sc[0] = ff; });
    ex.sync();
    SA_TICK(0);
    // covariance / rhs update: lane = rows r, r+64; wave w takes the column groups 8w, 8w+32, ..
    ex.par([&](int l) {
      const int w = l >> 6, r = l & 63;
      const double ffl = sc[0];
      double *dump = L.dump;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int row = r + 64 * h;
        if (row < no) {
          const double xi = L.X[row];
          for (int j = 8 * w; j < no; j += 8 * PW) {
            double m[8], xj[8];
            int e[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const int jj = j + u < no ? j + u : no - 1;
              e[u] = tri_off(no, jj) + (row - jj); xj[u] = L.X[jj]; m[u] = L.M[e[u] < 0 ? 0 : e[u]];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const double v = fma(lambda, m[u], ffl * (xi * xj[u]));
              double *dst = (row >= j + u && j + u < no) ? &L.M[e[u]] : dump;
              *dst = v;
            }
          }
          if (w == 0) breg[l].v[h] = fma(lambda, breg[l].v[h], ff * (xi * val));
        }
      }
    });
    SA_TICK(1);
    km++;
    if (km >= p.k) {
      km = 0;
      ex.par([&](int l) {
        for (int e = l; e < NMAX + kOlsPad; e += NL) L.Dv[e] = 0.0;
        if (l == 0) sc[1] = 1.0;
      });
      ex.sync();                                   // M complete, D cleared
      bool ok = true;
      ex.leader_par([&](int l) { sreg[l].v[0] = breg[l].v[0]; sreg[l].v[1] = breg[l].v[1]; });   // forward substitution starts from b
      for (int p4 = 0; p4 < no; p4 += PW) {
        const int nchunk = (p4 + 7) >> 3;           // chunks cover k < 8*nchunk <= NMAX; k >= p4 is masked by D == 0
        ex.par([&](int l) {
          const int w = l >> 6, r = l & 63;
          const int j = p4 + w;
          if (j < no) {
            double s_[2];
            const double *pa[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int row = r + 64 * h;
              const int rm = row < no ? row : no - 1;
              s_[h] = L.M[tri_off(no, j) + (rm - j) < 0 ? 0 : tri_off(no, j) + (rm - j)];
              if (row == j) s_[h] = s_[h] + nu;
              pa[h] = L.Lq + (row < S ? row : S - 1);   // own row: element k at pa[k*S]
            }
            const double *pb = L.Lq + j;              // row j
            struct Fc { double a0[8], a1[8], b[8], d[8]; };
            auto ld = [&](Fc &c, int m) {
              const int k = 8 * m;
#pragma unroll
              for (int u = 0; u < 8; u++) { c.a0[u] = pa[0][(k + u) * S]; c.a1[u] = pa[1][(k + u) * S]; c.b[u] = pb[(k + u) * S]; c.d[u] = L.Dv[k + u]; }
            };
            auto ac = [&](const Fc &c) {
              double t0[8], t1[8];
#pragma unroll
              for (int u = 0; u < 8; u++) { t0[u] = c.a0[u] * c.b[u]; t1[u] = c.a1[u] * c.b[u]; }
#pragma unroll
              for (int u = 0; u < 8; u++) { t0[u] = t0[u] * c.d[u]; t1[u] = t1[u] * c.d[u]; }
#pragma unroll
              for (int u = 0; u < 8; u++) { s_[0] = s_[0] - t0[u]; s_[1] = s_[1] - t1[u]; }
            };
            if (nchunk > 0) {
              Fc A, B;
              ld(A, 0);
              int m = 0;
              while (true) {
                if (m + 1 < nchunk) ld(B, m + 1);
                ac(A);
                if (++m >= nchunk) break;
                if (m + 1 < nchunk) ld(A, m + 1);
                ac(B);
                if (++m >= nchunk) break;
              }
            }
            ACC[(w * 2 + 0) * 64 + r] = s_[0];
            ACC[(w * 2 + 1) * 64 + r] = s_[1];
          }
        });
        ex.sync();
        if (ex.is_leader()) {
          // phase 2 (wave 0): right-looking inside the panel
          ex.leader_par([&](int l) {
#pragma unroll
            for (int c = 0; c < 2 * PW; c++) accp[l].v[c] = ACC[c * 64 + l];
          });
#pragma unroll
          for (int c = 0; c < PW; c++) {
            const int j = p4 + c;
            if (j >= no || !ok) break;
            const int hj = j >> 6, lj = j & 63;
            ex.leader_par([&](int l) { accc[l].v[0] = accp[l].v[2 * c]; accc[l].v[1] = accp[l].v[2 * c + 1]; tmpb[l] = hj ? accc[l].v[1] : accc[l].v[0]; });
            const double dj = ex.lane_bcast(tmpb, lj);
            if (dj < 1e-12) { ok = false; break; }
            const double invd = 1.0 / dj;
            ex.leader_par([&](int l) { tmpb[l] = hj ? sreg[l].v[1] : sreg[l].v[0]; });
            const double yk = ex.lane_bcast(tmpb, lj);     // forward substitution rides along (see ols_stage_fast)
            ex.leader_par([&](int l) {
#pragma unroll
              for (int h = 0; h < 2; h++) {
                const int row = l + 64 * h;
                const double v = accc[l].v[h] * invd;
                lpc[l].v[h] = v;
                if (row > j && row < no) L.Lq[j * S + row] = v;
                if (row == j) invd_mine[l].v[h] = invd;
                const double f = fold_fused(j, row) ? fma(-v, yk, sreg[l].v[h]) : sreg[l].v[h] - v * yk;
                if (row > j && row < no) sreg[l].v[h] = f;
              }
              if (l == 0) L.Dv[j] = dj;
            });
#pragma unroll
            for (int c2 = c + 1; c2 < PW; c2++) {
              const int j2 = p4 + c2;
              const int jb = j2 < NMAX ? j2 : NMAX - 1;
              const int hb = jb >> 6;
              ex.leader_par([&](int l) { tmpb[l] = hb ? lpc[l].v[1] : lpc[l].v[0]; });
              const double bq = ex.lane_bcast(tmpb, jb & 63);
              const bool fz = (c2 == c + 1) && (j2 & 1);          // k = j2-1 with j2 odd: the fused last term
              ex.leader_par([&](int l) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                  const double tt = lpc[l].v[h] * bq;
                  accp[l].v[2 * c2 + h] = fz ? fma(-tt, dj, accp[l].v[2 * c2 + h]) : accp[l].v[2 * c2 + h] - tt * dj;
                }
              });
            }
          }
          ex.leader_par([&](int l) { if (l == 0 && !ok) sc[1] = 0.0; });
        }
        ex.sync();
        ok = sc[1] != 0.0;
        if (!ok) break;
      }
      SA_TICK(2);
      if (ok && ex.is_leader()) {
        // (the forward substitution was carried along by the factorisation)
        ex.leader_par([&](int l) { zreg[l].v[0] = sreg[l].v[0] * invd_mine[l].v[0]; zreg[l].v[1] = sreg[l].v[1] * invd_mine[l].v[1]; });
        SA_TICK(3);
        ex.leader_par([&](int l) {
#pragma unroll
          for (int h = 0; h < 2; h++) { const int row = l + 64 * h; if (row < no) L.Dv[row] = zreg[l].v[h]; }     // z into LDS (D is no longer needed)
        });
        ex.wsync();
        ex.lane0([&]() {
          double wr[NMAX];
          const double *lb = L.Lq - (NMAX - no) * (S + 1);
          const double *zb = L.Dv - (NMAX - no);
          double *wb = L.Wv - (NMAX - no);
          OlsBwdRows<NMAX, S, 0>::run(no, L.X, lb, zb, wb, wr);
        });
        ex.wsync();
        SA_TICK(4);
      }
    }
    if (dec && t + 1 < n) {
      if (!sa_wait_ge(dec->prog_other, ols_other_need(p, n, t + 1), dec->fail)) return;
      ex.par([&](int l) {
        if (l < 64) {
#pragma unroll
          for (int h = 0; h < 2; h++) { const int row = l + 64 * h; if (row < no) xnext[l].v[h] = ols_x(p, self, other, n, t + 1, row); }
        }
      });
    }
    ex.sync();
    SA_TICK(5);
  }
  if (prof) ex.par([&](int l) { if (l == 0) for (int i = 0; i < 8; i++) prof[i] = tp[i]; });
#undef SA_TICK
}

SA_HD size_t ols_panel2_lds_bytes(int nmax) { return OlsLdsFast::bytes(nmax) + (size_t)(4 * 2 * 64 + 4) * sizeof(double); }

SA_HD size_t ols_panel_lds_bytes(int nmax, int waves) { return OlsLdsFast::bytes(nmax) + (size_t)(waves * 64 + 4) * sizeof(double); }

}  // namespace sacamd
