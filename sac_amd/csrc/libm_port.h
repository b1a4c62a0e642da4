// sac_amd/csrc/libm_port.h -- exp() and pow() with results bit-identical to the glibc 2.35
// x86-64 FMA variants (__exp_fma / __pow_fma) that the reference binary calls.
//
// The reference's losslessness rests on encoder and decoder computing the same fp64 values
// (SURVEY.md §7 hard part 1); its OLS stage amplifies a 1-ulp difference of pow() to ~1e-11
// relative in the prediction.  So the device code does not use the ROCm OCML functions but this
// port of the published algorithm (Szabolcs Nagy's exp/pow, glibc sysdeps/ieee754/dbl-64/
// e_exp.c, e_pow.c == ARM optimized-routines math/exp.c, pow.c), with the fused multiply-adds
// exactly where libm's compiled code has them (established from the disassembly).
// tests/test_emu_kernels.py checks both functions against the host libm on millions of inputs.
// Domain: pow(x>0 finite normal, y finite); exp(any finite).
#pragma once
#include "libm_tables.h"
#include "simt.h"

namespace sacamd {

#if defined(__HIP_DEVICE_COMPILE__)
#define SA_TABLE_QUAL __device__ const
#else
#define SA_TABLE_QUAL static const
#endif
SA_TABLE_QUAL double kPowPoly[7] = SA_POW_POLY;
SA_TABLE_QUAL double kPowLogTab[128 * 3] = SA_POW_LOGTAB;
SA_TABLE_QUAL double kExpPoly[4] = SA_EXP_POLY;
SA_TABLE_QUAL unsigned long long kExpTab[256] = SA_EXP_TAB;

SA_HD unsigned long long sa_asu(double x) { unsigned long long u; memcpy(&u, &x, 8); return u; }
SA_HD double sa_asd(unsigned long long u) { double x; memcpy(&x, &u, 8); return x; }

SA_HD double sa_exp_special(double tmp, unsigned long long sbits, unsigned long long ki) {
  if ((ki & 0x80000000ULL) == 0) {
    sbits -= 1009ULL << 52;
    const double scale = sa_asd(sbits);
    return 0x1p1009 * fma(scale, tmp, scale);
  }
  sbits += 1022ULL << 52;
  const double scale = sa_asd(sbits);
  const double prod = scale * tmp;
  double y = scale + prod;
  if (y < 1.0) {
    double lo = scale - y + prod;
    const double hi = 1.0 + y;
    lo = 1.0 - hi + y + lo;
    y = (hi + lo) - 1.0;
    if (y == 0.0) y = 0.0;
  }
  return 0x1p-1022 * y;
}

// shared core of exp(x) and pow's exp_inline(x, xtail)
SA_HD double sa_exp_core(double x, double xtail, bool standalone, const unsigned long long *exptab) {
  unsigned abstop = (unsigned)(sa_asu(x) >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
    if (abstop >= 0x409u) {
      if (standalone) {
        if (sa_asu(x) == 0xfff0000000000000ULL) return 0.0;
        if (abstop >= 0x7ffu) return 1.0 + x;
      }
      return (sa_asu(x) >> 63) ? 0.0 : sa_asd(0x7ff0000000000000ULL);
    }
    abstop = 0;
  }
  double kd = fma(SA_EXP_INVLN2N, x, SA_EXP_SHIFT);
  const unsigned long long ki = sa_asu(kd);
  kd -= SA_EXP_SHIFT;
  double r = fma(kd, SA_EXP_NEGLN2LON, fma(kd, SA_EXP_NEGLN2HIN, x));
  if (!standalone) r += xtail;
  const unsigned long long idx = 2 * (ki % 128);
  const unsigned long long top = ki << (52 - 7);
  const double tail = sa_asd(exptab[idx]);
  const unsigned long long sbits = exptab[idx + 1] + top;
  const double r2 = r * r;
  const double tmp = fma(r2 * r2, fma(r, kExpPoly[3], kExpPoly[2]), fma(r2, fma(r, kExpPoly[1], kExpPoly[0]), tail + r));
  if (abstop == 0) return sa_exp_special(tmp, sbits, ki);
  const double scale = sa_asd(sbits);
  return fma(scale, tmp, scale);
}

SA_HD double sa_exp(double x) { return sa_exp_core(x, 0.0, true, kExpTab); }
// same, with the 2 KB table staged by the caller (LDS)
SA_HD double sa_exp_t(double x, const unsigned long long *exptab) { return sa_exp_core(x, 0.0, true, exptab); }

// logtab: 128 x {invc, logc, logctail}; exptab: 256 words
SA_HD double sa_pow_t(double x, double y, const double *logtab, const unsigned long long *exptab) {
  const unsigned long long ix = sa_asu(x), iy = sa_asu(y);
  const unsigned topy = (unsigned)(iy >> 52) & 0x7ff;
  if (2 * iy == 0) return 1.0;
  if (ix == 0x3ff0000000000000ULL) return 1.0;
  if (topy - 0x3beu >= 0x43eu - 0x3beu) {
    if (topy < 0x3beu) return 1.0;                      // |y| < 2^-65
    const bool big = (ix > 0x3ff0000000000000ULL) == !(iy >> 63);   // |y| >= 2^63
    return big ? sa_asd(0x7ff0000000000000ULL) : 0.0;
  }
  // log_inline
  const unsigned long long tmp = ix - 0x3fe6955500000000ULL;
  const int i = (int)((tmp >> (52 - 7)) % 128);
  const int k = (int)((long long)tmp >> 52);
  const unsigned long long iz = ix - (tmp & (0xfffULL << 52));
  const double z = sa_asd(iz), kd = (double)k;
  const double invc = logtab[3 * i], logc = logtab[3 * i + 1], logctail = logtab[3 * i + 2];
  const double r = fma(z, invc, -1.0);
  const double t1 = fma(kd, SA_POW_LN2HI, logc);
  const double t2 = t1 + r;
  const double lo1 = fma(kd, SA_POW_LN2LO, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = kPowPoly[0] * r, ar2 = r * ar, ar3 = r * ar2;
  const double hi = t2 + ar2;
  const double lo3 = fma(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double q = fma(ar2, fma(ar2, fma(r, kPowPoly[6], kPowPoly[5]), fma(r, kPowPoly[4], kPowPoly[3])), fma(r, kPowPoly[2], kPowPoly[1]));
  const double lo = fma(ar3, q, lo1 + lo2 + lo3 + lo4);
  const double yl = hi + lo;
  const double tl = hi - yl + lo;
  const double ehi = y * yl;
  const double elo = fma(y, tl, fma(y, yl, -ehi));
  return sa_exp_core(ehi, elo, false, exptab);
}
SA_HD double sa_pow(double x, double y) { return sa_pow_t(x, y, kPowLogTab, kExpTab); }

constexpr int kLibmLdsDoubles = 128 * 3 + 256;   // log table + exp table staged in LDS
// stage both tables into LDS (or any buffer); lane l of nl
SA_HD void sa_stage_tables(double *dst, int l, int nl) {
  for (int i = l; i < 128 * 3; i += nl) dst[i] = kPowLogTab[i];
  unsigned long long *e = reinterpret_cast<unsigned long long *>(dst + 128 * 3);
  for (int i = l; i < 256; i += nl) e[i] = kExpTab[i];
}

}  // namespace sacamd
