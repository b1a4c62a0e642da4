// sac_amd/csrc/canon.h -- the "canonical arithmetic" of the reference binary, as host/device
// inline functions with every fused multiply-add explicit (compile with -ffp-contract=off).
//
// These follow slmath::dot / std::transform_reduce / the vectorised fold-left reduction loops as
// g++ 11 -O3 -mavx2 -mfma compiles /root/reference/src/common/math.h:14-191; DESIGN.md
// ("canonical arithmetic") explains how the forms were established.  They are used wherever the
// product must agree with the reference decoder to the last bit (OLS stage, small dots of the
// RLS / mixer / bias chain).
#pragma once
#include "simt.h"

namespace sacamd {

// in-order reduction acc += a[k]*b[k]: groups of 4 and one pair unfused, odd last element fused
template <class PA, class PB>
SA_HD double fold_add(double acc, int m, PA a, PB b) {
  int k = 0;
  for (; k + 4 <= m; k += 4) {
    acc = acc + a(k) * b(k);
    acc = acc + a(k + 1) * b(k + 1);
    acc = acc + a(k + 2) * b(k + 2);
    acc = acc + a(k + 3) * b(k + 3);
  }
  if (m - k >= 2) {
    acc = acc + a(k) * b(k);
    acc = acc + a(k + 1) * b(k + 1);
    k += 2;
  }
  if (k < m) acc = fma(a(k), b(k), acc);
  return acc;
}

// true when term k of an m-term fold-left chain is the fused one
SA_HD bool fold_fused(int k, int m) { return (m & 1) && (k == m - 1); }

// std::transform_reduce(x, x+n, y, 0.0) as compiled
SA_HD double tr_dot(const double *a, const double *b, int n) {
  double init = 0.0;
  while (n >= 4) {
    const double v1 = fma(a[1], b[1], a[0] * b[0]);
    const double v2 = fma(a[3], b[3], a[2] * b[2]);
    init = init + (v1 + v2);
    a += 4; b += 4; n -= 4;
  }
  return fold_add(init, n, [&](int k) { return a[k]; }, [&](int k) { return b[k]; });
}

// slmath::dot (common/math.h:130-161)
SA_HD double dot_canon(const double *x, const double *y, int n) {
  double total = 0.0;
  int i = 0;
  if (n >= 8) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (; i + 8 <= n; i += 8) {
      s0 = fma(x[i], y[i], s0); s1 = fma(x[i + 1], y[i + 1], s1);
      s2 = fma(x[i + 2], y[i + 2], s2); s3 = fma(x[i + 3], y[i + 3], s3);
      t0 = fma(x[i + 4], y[i + 4], t0); t1 = fma(x[i + 5], y[i + 5], t1);
      t2 = fma(x[i + 6], y[i + 6], t2); t3 = fma(x[i + 7], y[i + 7], t3);
    }
    s0 = s0 + t0; s1 = s1 + t1; s2 = s2 + t2; s3 = s3 + t3;
    total = ((s0 + s1) + s2) + s3;
  }
  total += tr_dot(x + i, y + i, n - i);
  return total;
}

// dot_canon with a compile-time length (first operand may live in registers)
template <int N, class A, class B>
SA_HD double dot_canon_n(A a, B b) {
  double total = 0.0;
  constexpr int blocks = (N >= 8) ? N / 8 : 0;
  if (blocks) {
    double s[4] = {0, 0, 0, 0}, t[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < blocks * 8; i += 8) {
#pragma unroll
      for (int c = 0; c < 4; c++) { s[c] = fma(a(i + c), b(i + c), s[c]); t[c] = fma(a(i + 4 + c), b(i + 4 + c), t[c]); }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) s[c] = s[c] + t[c];
    total = ((s[0] + s[1]) + s[2]) + s[3];
  }
  constexpr int i0 = blocks * 8;
  double init = 0.0;
  constexpr int rem = N - i0;
  constexpr int g = (rem >= 4) ? 4 : 0;
  if (g) {
    const double v1 = fma(a(i0 + 1), b(i0 + 1), a(i0) * b(i0));
    const double v2 = fma(a(i0 + 3), b(i0 + 3), a(i0 + 2) * b(i0 + 2));
    init = init + (v1 + v2);
  }
  constexpr int j0 = i0 + g, m = N - j0;
  int k = 0;
  if (m >= 2) { init = init + a(j0) * b(j0); init = init + a(j0 + 1) * b(j0 + 1); k = 2; }
  if (k < m) init = fma(a(j0 + k), b(j0 + k), init);
  total += init;
  return total;
}

// the part of slmath::calc_s2pow (common/math.h:164-191) that std::transform_reduce handles: the < 4 elements
// after the AVX2 chains, or all n < 8 elements of a short vector; terms are pw * (x * x)
SA_HD double tr_s2pow(const double *x, const double *pw, int n) {
  double init = 0.0;
  while (n >= 4) {
    const double v1 = fma(x[1] * x[1], pw[1], (x[0] * x[0]) * pw[0]);
    const double v2 = fma(x[3] * x[3], pw[3], (x[2] * x[2]) * pw[2]);
    init = init + (v1 + v2);
    x += 4; pw += 4; n -= 4;
  }
  return fold_add(init, n, [&](int k) { return x[k] * x[k]; }, [&](int k) { return pw[k]; });
}

SA_HD double sgnd(double x) { return (double)((x > 0) - (x < 0)); }
SA_HD double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
SA_HD int clampi32(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// (int32_t)double as the reference's x86-64 build performs it (cvttsd2si): a value outside the int32 range -- or a NaN --
// gives INT_MIN ("integer indefinite"), where gfx950's v_cvt_i32_f64 saturates to INT_MAX on the positive side.  It matters:
// on 24-bit material the cascade overshoots past 2^31 in the first samples of a frame (libsac.cpp:106 then clamps INT_MIN to
// the frame's minimum); found on the GPU in round 4 (profiles/r04/bisect24_stages.log).
SA_HD int cvt_i32_x86(double r) { return (r >= -2147483648.0 && r < 2147483648.0) ? (int)r : (-2147483647 - 1); }

}  // namespace sacamd
