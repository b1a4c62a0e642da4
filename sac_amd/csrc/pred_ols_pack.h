// sac_amd/csrc/pred_ols_pack.h -- stage 1 (OLS) for short regressors, SEVERAL work-items per wave (round 4).
//
// Reference: OLS (/root/reference/src/pred/ols.cpp:7-57) + slmath::LDLT (common/math.h:14-78) + RunSumGEO
// (common/utils.h:39-72), regressor of Predictor::fillbuf_ch0/ch1 (libsac/pred.cpp:17-31) -- the arithmetic of
// ols_stage_reg (pred_ols.h), element for element and in the same order, so p_lpc stays bit-identical.
//
// Why: the one-wave kernel gives a matrix ROW to a lane, so a 16-tap item keeps 16 of 64 lanes busy and a 24- / 32-tap
// item half of them, and the stage is bound by instruction issue (a wave-wide fp64 instruction costs its four cycles
// whether 1 or 64 lanes do something useful).  Here a wave is cut into G = 64 / GL groups of GL lanes (GL = 16: four
// items per wave, GL = 32: two) and lane l works on row l % GL of item l / GL: one instruction stream, G factorisations.
//   * every per-item scalar of ols_stage_reg (prediction, forgetting factor, pivot, ...) is a per-lane value that the GL
//     lanes of a group hold alike; a group's regressor, weights, L and z live in the group's own LDS block and are read
//     back as broadcasts (address = group base + uniform offset);
//   * the pivot D[k] and the forward-substitution value y[k] come from lane k of EACH group (grp_bcast: ds_bpermute);
//   * the column loop runs to the longest regressor of the wave; a group whose regressor is shorter only masks its
//     stores (what it computes beyond its last column is never read), a group whose pivot falls below 1e-12 keeps its old
//     weights (math.h:36-37, ols.cpp:49-50) while the others go on;
//   * the serial back-substitution (n^2 / 2 dependent FMAs) runs on lane 0 of every group at once.
// All work-items of a wave share the solve interval k (search: optk, final pass: 1).
#pragma once
#include "pred_ols.h"

namespace sacamd {

struct OlsPackSlot {          // one work-item of a packed wave (group-uniform); n == 0: empty group
  const ChanParam *p;
  const int *self, *other;
  double *out;                // p_lpc [n]
  int n;
};

template <int NMAX> SA_HD constexpr int ols_pack_group_doubles() { return NMAX + 8 + 8 + 2 * (NMAX + kOlsPad) + NMAX * (NMAX + kOlsPad); }
template <int NMAX, int GL> SA_HD constexpr size_t ols_pack_lds_bytes() {
  return (size_t)(kLibmLdsDoubles + (64 / GL) * ols_pack_group_doubles<NMAX>()) * sizeof(double) + 16;
}

// slots Q .. NMAX-2 of one column step, as OlsRankOne (pred_ols.h) but with the column's L[j][k] per lane (each group reads
// its own column): V[q] = V[q+1] - (lk * L[k+1+q][k]) * dk, the first slot fused when k is even (canon.h fold_add)
template <int NMAX, int Q>
struct OlsRankOneP {
  template <class E, class RV, class RD, class RB, class RI>
  static SA_HD __attribute__((always_inline)) void run(E &ex, int rem /* uniform: columns after k of the longest group */, const double *lds, const RI &lrow_off,
                                                       bool k_even, RV &V, const RD &lk, const RD &dk, RB &bj) {
    if constexpr (Q < NMAX - 1) {
      if (Q < rem) {
        constexpr int QE = (Q + 4) < (NMAX - 1) ? (Q + 4) : (NMAX - 1);
        ex.par([&](int l) {
          double bn[4] = {0.0, 0.0, 0.0, 0.0};
          if constexpr (QE < NMAX - 1) {
            const double *lrow = lds + lrow_off[l];
#pragma unroll
            for (int u = 0; u < 4; u++) bn[u] = lrow[QE + u];       // the next group of slots: its LDS round trip runs under this group's arithmetic
          }
          double tt[4];
#pragma unroll
          for (int u = 0; u < 4; u++) if (Q + u < QE) tt[u] = lk[l] * bj[l].v[u];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (Q + u < QE) {
              if (Q + u == 0) V[l].v[0] = k_even ? fma(-tt[0], dk[l], V[l].v[1]) : V[l].v[1] - tt[0] * dk[l];
              else V[l].v[Q + u] = V[l].v[Q + u + 1] - tt[u] * dk[l];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) if (Q + u < QE) SA_PIN_F64(V[l].v[Q + u]);
#pragma unroll
          for (int u = 0; u < 4; u++) bj[l].v[u] = bn[u];
        });
        OlsRankOneP<NMAX, QE>::run(ex, rem, lds, lrow_off, k_even, V, lk, dk, bj);
      }
    }
  }
};

// covariance update of columns J .. NMAX-1 (ols.cpp:38-42): M[i][j] = lambda M[i][j] + ff (x_i x_j), x_j read from the group's X
template <int NMAX, int J>
struct OlsCovUpdateP {
  template <class E, class RV, class RD, class RI>
  static SA_HD __attribute__((always_inline)) void run(E &ex, int nomax, const double *lds, const RI &x_off, RV &M, const RD &xr, const RD &lambda, const RD &ff) {
    if constexpr (J < NMAX) {
      if constexpr ((J & 7) == 0) SA_OPAQUE_SINT(nomax);
      if (J < nomax) {
        constexpr int JE = (J + 4) < NMAX ? (J + 4) : NMAX;
        ex.par([&](int l) {
          const double *X = lds + x_off[l];
          double xj[4];
#pragma unroll
          for (int u = 0; u < 4; u++) if (J + u < JE) xj[u] = X[J + u];       // columns beyond a group's regressor read its zero padding
#pragma unroll
          for (int u = 0; u < 4; u++) if (J + u < JE) M[l].v[J + u] = fma(lambda[l], M[l].v[J + u], ff[l] * (xr[l] * xj[u]));
#pragma unroll
          for (int u = 0; u < 4; u++) if (J + u < JE) SA_PIN_F64(M[l].v[J + u]);
        });
        OlsCovUpdateP<NMAX, JE>::run(ex, nomax, lds, x_off, M, xr, lambda, ff);
      }
    }
  }
};

// rows ip = IP .. of the back-substitution (math.h:67-73), row i = no - 1 - ip of every group at once (lane 0 of the group):
// w[i] = z[i] - sum_{k = i+1 .. no-1} L[k][i] w[k], a fused chain in ascending k; L[k][i] for k = i+1.. are consecutive in
// column i of Lq, w[k] is wr[no-1-k] (compile-time index).  Lanes whose group has fewer rows keep computing on row 0's data
// (in range, never stored).
template <int NMAX, int S, int IP>
struct OlsBwdRowsP {
  static SA_HD __attribute__((always_inline)) void run(int nomax, int no, const double *Lq, const double *Dv, double *Wv, double (&wr)[NMAX]) {
    if constexpr (IP < NMAX) {
      if (IP >= nomax) return;
      const int i = no - 1 - IP;
      const bool valid = i >= 0;
      int ic = valid ? i : 0;
      SA_OPAQUE_INT(ic);                                    // one address per row; the row's elements at immediate offsets
      double s_ = Dv[ic];
      const double *rowp = Lq + ic * (S + 1) + 1;
      // the row's L elements are requested CH at a time, one chunk ahead of the chain (the compiler would otherwise hoist all
      // loads of the unrolled rows and run out of registers)
      constexpr int CH = 8, NCH = (IP + CH - 1) / CH;
      double buf[2][CH];
#pragma unroll
      for (int q = 0; q < CH; q++) if (q < IP) buf[0][q] = rowp[q];
#pragma unroll
      for (int c = 0; c < NCH; c++) {
#pragma unroll
        for (int q = 0; q < CH; q++) if ((c + 1) * CH + q < IP) buf[(c + 1) & 1][q] = rowp[(c + 1) * CH + q];
#pragma unroll
        for (int q = 0; q < CH; q++) if (c * CH + q < IP) s_ = fma(-buf[c & 1][q], wr[IP - 1 - (c * CH + q)], s_);
#pragma unroll
        for (int q = 0; q < CH; q++) if ((c + 1) * CH + q < IP) SA_PIN_F64(buf[(c + 1) & 1][q]);
      }
      wr[IP] = s_;
      if (valid) Wv[ic] = s_;
      OlsBwdRowsP<NMAX, S, IP + 1>::run(nomax, no, Lq, Dv, Wv, wr);
    }
  }
};

template <class E, int NMAX, int GL>
SA_HD void ols_stage_pack(E &ex, const typename E::template Reg<OlsPackSlot> &slot, int kk /* solve interval, the same for every item */, char *lds_base) {
  static_assert(E::nl == 64 && (GL == 16 || GL == 32) && NMAX <= GL && NMAX % 4 == 0 && NMAX <= 32, "packed one-wave path");
  constexpr int G = 64 / GL, S = NMAX + kOlsPad, NL = 64;
  constexpr int kGrp = ols_pack_group_doubles<NMAX>();
  constexpr int oC = NMAX, oP = NMAX + 8, oW = NMAX + 16, oD = oW + NMAX + kOlsPad, oL = oD + NMAX + kOlsPad;   // X | C (dot accumulators) | P (item constants) | Wv | Dv | Lq
  double *lds = reinterpret_cast<double *>(lds_base);
  double *libm = lds;
  const unsigned long long *exptab = reinterpret_cast<const unsigned long long *>(libm + 128 * 3);

  typename E::template Reg<double> xr, breg, sreg, zreg, invd_mine, lk, dk, yk, ff, esum, lam, valr;   // (the item's other constants are read from its LDS block where needed)
  typename E::template Reg<OlsRow<NMAX>> M, V;
  typename E::template Reg<DArr4> bj;
  typename E::template Reg<int> xnext, no, nn, pa, pb, pdu, x_off, lrow_off, okr;
  typename E::template Reg<const int *> selfp, otherp;
  typename E::template Reg<double *> outp;

  auto regx = [&](int l, int t, int j) -> int {      // ols_x (pred_ols.h) on the lane's own item
    if (j < pa[l]) { const int i = t - pa[l] + j; return i >= 0 ? selfp[l][i] : 0; }
    int u = t - pdu[l]; if (u < 0) u = 0;
    const int i = u - pb[l] + (j - pa[l]);
    return (i >= 0 && i < nn[l]) ? otherp[l][i] : 0;
  };

  ex.par([&](int l) {
    const OlsPackSlot &sl = slot[l];
    const int g = l / GL, r = l % GL;
    const ChanParam &p = *sl.p;
    const bool on = sl.n > 0;
    nn[l] = on ? sl.n : 0; no[l] = on ? p.n_ols : 0;
    pa[l] = p.a; pb[l] = p.b; pdu[l] = p.du;
    selfp[l] = sl.self; otherp[l] = sl.other; outp[l] = sl.out;
    lam[l] = p.lambda;
    x_off[l] = kLibmLdsDoubles + g * kGrp; lrow_off[l] = 0; okr[l] = 0;
    xr[l] = 0.0; breg[l] = 0.0; sreg[l] = 0.0; zreg[l] = 0.0; invd_mine[l] = 0.0; lk[l] = 0.0; dk[l] = 0.0; yk[l] = 0.0;
    ff[l] = 0.0; esum[l] = 0.0; valr[l] = 0.0;
#pragma unroll
    for (int j = 0; j < NMAX; j++) { M[l].v[j] = 0.0; V[l].v[j] = 0.0; }
#pragma unroll
    for (int u = 0; u < 4; u++) bj[l].v[u] = 0.0;
    double *gb = lds + x_off[l];
    for (int e = r; e < kGrp; e += GL) gb[e] = 0.0;          // X, C, Wv, Dv, Lq: all zero (padding rows stay zero for ever)
    sa_stage_tables(libm, l, NL);
    const int row = r;
    xnext[l] = (row < no[l] && nn[l] > 0) ? regx(l, 0, row) : 0;
  });
  ex.sync();
  ex.par([&](int l) {
    if (l % GL == 0) { const ChanParam &p = *slot[l].p; double *P = lds + x_off[l] + oP; P[0] = p.nu_eff; P[1] = p.beta_sum; P[2] = p.beta_pow; P[3] = p.beta_add; }
  });
  ex.sync();

  int nmax = 0, nomax = 0;
  for (int g = 0; g < G; g++) {
    const int a = ex.lane_geti(nn, g * GL), b = ex.lane_geti(no, g * GL);
    nmax = a > nmax ? a : nmax; nomax = b > nomax ? b : nomax;
  }
  nmax = E::uniform(nmax); nomax = E::uniform(nomax); kk = E::uniform(kk);
  int km = 0;
  for (int t = 0; t < nmax; t++) {
    // ---- regressor of this step into registers and the group's X; next step's loads are issued now
    ex.par([&](int l) {
      const int r = l % GL;
      xr[l] = (double)xnext[l];
      if (r < no[l]) lds[x_off[l] + r] = xr[l];
      const int tn = t + 1 < nn[l] ? t + 1 : nn[l] - 1;          // a group past its last sample idles on that sample
      xnext[l] = (r < no[l] && nn[l] > 0) ? regx(l, tn, r) : 0;
      const int tc = t < nn[l] ? t : nn[l] - 1;
      valr[l] = nn[l] > 0 ? (double)selfp[l][tc] : 0.0;
    });
    ex.sync();
    // ---- prediction = slmath::dot(x, w) (math.h:130-161): eight FMA accumulators on lanes r = 0..7 of the group ...
    ex.par([&](int l) {
      const int r = l % GL;
      if (r < 8) {
        const double *X = lds + x_off[l], *W = X + oW;
        double c = 0.0;
        for (int i = 0; i + 8 <= no[l]; i += 8) c = fma(X[i + r], W[i + r], c);
        lds[x_off[l] + oC + r] = c;
      }
    });
    ex.wsync();
    // ... (s_c + t_c), ((s0+s1)+s2)+s3, the transform_reduce tail; then the IRLS weight of the step (ols.cpp:27-36)
    ex.par([&](int l) {
      const double *X = lds + x_off[l], *W = X + oW, *C = X + oC;
      const int n_ = no[l], nb = n_ & ~7;
      double total = 0.0;
      if (nb) {
        const double s0 = C[0] + C[4], s1 = C[1] + C[5], s2 = C[2] + C[6], s3 = C[3] + C[7];
        total = ((s0 + s1) + s2) + s3;
      }
      total += tr_dot(X + nb, W + nb, n_ - nb);
      if (l % GL == 0 && t < nn[l]) outp[l][t] = total;
      const double *P = X + oP;
      const double e = valr[l] - total;
      esum[l] = fma(P[1], esum[l], fabs(e));
      const double c = sa_pow_t(esum[l] + P[3], -P[2], libm, exptab);
      ff[l] = (1.0 - lam[l]) * c;
    });
    // ---- covariance and right-hand side (ols.cpp:38-45)
    OlsCovUpdateP<NMAX, 0>::run(ex, nomax, lds, x_off, M, xr, lam, ff);
    ex.par([&](int l) { breg[l] = fma(lam[l], breg[l], ff[l] * (xr[l] * valr[l])); });
    km++;
    if (km >= kk) {
      km = 0;
      // ---- LDL^T of A + nu I, right-looking on register rows, forward substitution riding along (math.h:21-66)
      ex.par([&](int l) {
        const int r = l % GL;
        const double nu = lds[x_off[l] + oP];
#pragma unroll
        for (int j = 0; j < NMAX; j++) V[l].v[j] = (r == j) ? M[l].v[j] + nu : M[l].v[j];
        sreg[l] = breg[l];
        okr[l] = 1;
      });
      for (int k = 0; k < nomax; k++) {
        ex.template grp_bcast_col<GL>(dk, V, 0, k);            // pivot D[k] = V[k][k] (slot 0 holds column k) of every group
        ex.template grp_bcast<GL>(yk, sreg, k);                // y[k] is final
        ex.par([&](int l) {
          const int r = l % GL;
          const bool live = k < no[l];
          if (live && dk[l] < 1e-12) okr[l] = 0;               // LDLT::Factor fails (math.h:36-37): this group keeps its old weights
          const double invd = 1.0 / dk[l];
          const double lp = V[l].v[0] * invd;                  // L[i][k] = lij * invD (math.h:49)
          lk[l] = lp;
          double *Lq = lds + x_off[l] + oL;
          if (live && r > k && r < no[l]) Lq[k * S + r] = lp;
          if (r == k) invd_mine[l] = invd;
          const double v = fold_fused(k, r) ? fma(-lp, yk[l], sreg[l]) : sreg[l] - lp * yk[l];
          if (live && r > k && r < no[l]) sreg[l] = v;
          lrow_off[l] = x_off[l] + oL + k * S + k + 1;
        });
        ex.wsync();
        int rem = nomax - 1 - k;
        SA_OPAQUE_SINT(rem);
        ex.par([&](int l) {
          const double *lrow = lds + lrow_off[l];
#pragma unroll
          for (int u = 0; u < 4; u++) bj[l].v[u] = lrow[u];
        });
        OlsRankOneP<NMAX, 0>::run(ex, rem, lds, lrow_off, (k & 1) == 0, V, lk, dk, bj);
      }
      ex.par([&](int l) {
        const int r = l % GL;
        zreg[l] = sreg[l] * invd_mine[l];
        if (okr[l] && r < no[l]) lds[x_off[l] + oD + r] = zreg[l];
      });
      ex.wsync();
      // ---- back-substitution: lane 0 of every group (the groups' chains run side by side)
      ex.par([&](int l) {
        if (l % GL == 0 && okr[l]) {
          double wr[NMAX];
          const double *X = lds + x_off[l];
          OlsBwdRowsP<NMAX, S, 0>::run(nomax, no[l], X + oL, X + oD, lds + x_off[l] + oW, wr);
        }
      });
      ex.wsync();
    }
    ex.sync();
  }
}

}  // namespace sacamd
