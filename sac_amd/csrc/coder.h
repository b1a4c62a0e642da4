// sac_amd/csrc/coder.h -- bitplane / SSE context-mixing coder + binary range coder.
//
// Reference: BitplaneCoder (/root/reference/src/libsac/vle.{h,cpp}), LinearCounterLimit /
// LinearCounter16 (model/counter.h), NMixLogistic (model/mixer.h), SSENL<15> (model/sse.h),
// LogDomain (model/domain.h), RangeCoderSH (model/range.cpp:54-92), MapEncoder
// (libsac/map.cpp:3-101).  Integer arithmetic throughout -> bit-exact.
//
// One wave per stream (frame x channel [x variant]).  Everything the coder conditions on
// except its adaptive state is a pure function of the data: whether neighbour k is already
// "significant" when sample s is coded in plane b is (msbpos[k] >= b) for k < s and
// (msbpos[k] > b) for k >= s (vle.cpp:209-229).  So for each chunk of 64 samples the 64 lanes
// compute all context indices, the Laplace prior (table lookup) and the bit in parallel, and
// only the adaptive chain (counters -> mixer -> 2 x SSE -> final mix -> range coder -> updates)
// runs serially.  Model state lives in LDS; the 2^16-entry significance counter table and the
// 4.5 MB Laplace table (precomputed on the host with the reference's libm expressions,
// vle.cpp:70-79) live in HBM/L2.
#pragma once
#include "simt.h"

namespace sacamd {

constexpr int kPBits = 15, kPScale = 1 << 15, kPScaleM = kPScale - 1;
constexpr int kLaplaceAvg = 1 << 17;      // avg_sum domain of the host table
constexpr int kLaplacePlanes = 18;
constexpr int kCoderChunk = 64;
constexpr int kCoderHalo = 32;

struct CntL { unsigned short p1, cnt; };

struct CoderModel {
  CntL csig1[80], cref0[32], cref1[256], cref2[64], cref3[160], p_laplace[32];
  int lmixref[32][5], lmixsig[128][3], ssemix[2];
  unsigned short sse[160][2][16];
  unsigned char sse_lb[160];
};

struct CoderDesc {           // per-sample descriptor of one chunk (structure of arrays in LDS)
  unsigned short pest[kCoderChunk], i1[kCoderChunk], i2[kCoderChunk], i3[kCoderChunk], i4[kCoderChunk];
  unsigned char type[kCoderChunk], bit[kCoderChunk], mix[kCoderChunk], s1[kCoderChunk], s2[kCoderChunk];
};

struct CoderWin {            // staged data window of one chunk (with halo)
  int val[kCoderChunk + 2 * kCoderHalo];
  unsigned char msb[kCoderChunk + 2 * kCoderHalo];
};

struct RangeEnc {            // RangeCoderSH, encode side
  unsigned range, FFNum, Cache;
  unsigned long long lowc;
  unsigned char *out;
  int pos, cap;
  bool store;                // only the lane that owns the output stores
  SA_HD void init(unsigned char *o, int capacity, bool st) { range = 0xFFFFFFFFu; FFNum = 0; Cache = 0; lowc = 0; out = o; pos = 0; cap = capacity; store = st; }
  SA_HD void put(unsigned b) { if (store && pos < cap) out[pos] = (unsigned char)b; pos++; }
  SA_HD void shift_low() {
    const unsigned Carry = (unsigned)(lowc >> 32), low = (unsigned)lowc;
    if (low < 0xFF000000u || Carry) {
      put(Cache + Carry);
      for (; FFNum != 0; FFNum--) put(Carry - 1);
      Cache = low >> 24;
    } else FFNum++;
    lowc = (unsigned long long)(unsigned)(low << 8);
  }
  SA_HD void encode(unsigned p1, int bit) {
    const unsigned rnew = (unsigned)(((unsigned long long)range * ((unsigned)(kPScale - p1) << (32 - kPBits))) >> 32);
    if (bit) { range -= rnew; lowc += rnew; } else range = rnew;
    while (range < 0x01000000u) { range <<= 8; shift_low(); }
  }
  SA_HD void stop() { for (int i = 0; i < 5; i++) shift_low(); }
};

SA_HD int idiv_s(int val, int s) { return val < 0 ? -(((-val) + (1 << (s - 1))) >> s) : (val + (1 << (s - 1))) >> s; }
SA_HD int idiv_s64(long long val, int s) { return (int)(val < 0 ? -(((-val) + (1LL << (s - 1))) >> s) : (val + (1LL << (s - 1))) >> s); }
SA_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
SA_HD int ilog2i(int v) { int nb = 0; while (v >>= 1) nb++; return nb; }

SA_HD void cntl_update(CntL &c, int bit, int limit) {           // counter.h:58-68
  unsigned cnt = c.cnt;
  if ((int)cnt < limit) cnt++;
  const int d = kPScale / ((int)cnt + 3);
  const int p1 = c.p1;
  const int dp = bit ? ((kPScale - p1) * d) >> kPBits : -((p1 * d) >> kPBits);
  c.p1 = (unsigned short)clampi(p1 + dp, 1, kPScaleM);
  c.cnt = (unsigned short)cnt;
}
SA_HD void cnt16_update(unsigned short &p1, int bit, int L) {   // counter.h:31-37
  const int err = (bit << kPBits) - (int)p1;
  p1 = (unsigned short)clampi((int)p1 + idiv_s(L * err, kPBits), 1, kPScaleM);
}
SA_HD int squash(const unsigned short *inv, int x) { return x < -2047 ? 1 : (x > 2047 ? kPScaleM : (int)inv[x + 2047]); }

template <int N>
SA_HD int mix_predict(const int *w, const int *st, const unsigned short *inv) {   // mixer.h:76-87
  long long sum = 0;
#pragma unroll
  for (int i = 0; i < N; i++) sum += (long long)(w[i] * st[i]);
  return clampi(squash(inv, idiv_s64(sum, 16)), 1, kPScaleM);
}
template <int N>
SA_HD void mix_update(int *w, const int *st, int pd, int bit, int rate) {          // mixer.h:88-96
  const int err = (bit << kPBits) - pd;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int de = idiv_s(st[i] * err, 12);
    const int wd = idiv_s(de * rate, 12);
    w[i] = clampi(w[i] + wd, -(1 << 19), (1 << 19) - 1);
  }
}
// SSENL<N>::Predict (sse.h:101-113); returns prediction, *pq = quantised bin
template <int N>
SA_HD int sse_predict(const unsigned short *map /*[N+1]*/, int stp, int *pq) {
  constexpr int tscale = 2662, xscale = (2 * tscale) / (N - 1);
  int q = stp + tscale;
  q = q < 0 ? 0 : (q > 2 * tscale ? 2 * tscale : q);
  const int pquant = q / xscale, pmod = q - pquant * xscale;
  const int pl = map[pquant], ph = map[pquant + 1];
  *pq = pquant;
  return clampi((pl * (xscale - pmod) + ph * pmod) / xscale, 1, kPScaleM);
}

SA_HD void coder_model_init(CoderModel &m, const unsigned short *inv, const unsigned short *plap_init, int lane, int nl) {
  auto fill = [&](CntL *a, int n) { for (int i = lane; i < n; i += nl) { a[i].p1 = kPScale >> 1; a[i].cnt = 0; } };
  fill(m.csig1, 80); fill(m.cref0, 32); fill(m.cref1, 256); fill(m.cref2, 64); fill(m.cref3, 160);
  for (int i = lane; i < 32; i += nl) { m.p_laplace[i].p1 = plap_init[i]; m.p_laplace[i].cnt = 0; }
  for (int i = lane; i < 32 * 5; i += nl) (&m.lmixref[0][0])[i] = 0;
  for (int i = lane; i < 128 * 3; i += nl) (&m.lmixsig[0][0])[i] = 0;
  if (lane < 2) m.ssemix[lane] = 0;
  for (int i = lane; i < 160 * 2 * 16; i += nl) {
    const int k = i & 15;
    const int x = k * 380 - 2662;       // SSENL<15>: xscale 380, tscale 2662 (sse.h:91-99)
    (&m.sse[0][0][0])[i] = (unsigned short)squash(inv, x);
  }
  for (int i = lane; i < 160; i += nl) m.sse_lb[i] = 0;
}

// ---- per-sample context computation (data only).  W is the staged window of the chunk that
// starts at sample s0; local index of sample s is s - s0 + kCoderHalo.
SA_HD int msb_seen(const CoderWin &W, int li, bool before, int bpn) {
  const int m = W.msb[li];
  return before ? (m >= bpn ? m : 0) : (m > bpn ? m : 0);
}

SA_HD void coder_describe(const CoderWin &W, CoderDesc &D, int i, int s, int n, int bpn, const unsigned short *laplace) {
  const int li = i + kCoderHalo;
  // GetAvgSum(32), vle.cpp:54-68
  unsigned long long nsum = 0; int nidx = 0;
  const unsigned ml = ~((1u << bpn) - 1), mr = ~((1u << (bpn + 1)) - 1);
  for (int d = -32; d <= 32; d++) {
    const int k = s + d;
    if (k >= 0 && k < n) { nsum += (unsigned)W.val[li + d] & (d < 0 ? ml : mr); nidx++; }
  }
  const unsigned avg = nidx > 0 ? (unsigned)((nsum + (nidx - 1)) / nidx) : 0;
  const int pest = laplace[(size_t)bpn * kLaplaceAvg + (avg < (unsigned)kLaplaceAvg ? avg : (unsigned)kLaplaceAvg - 1)];
  // GetSigState, vle.cpp:33-52
  int sig[17];
  sig[0] = msb_seen(W, li, false, bpn);
  for (int d = 1; d <= 8; d++) {
    sig[2 * d - 1] = (s > d - 1) ? msb_seen(W, li - d, true, bpn) : 0;
    sig[2 * d] = (s < n - d) ? msb_seen(W, li + d, false, bpn) : 0;
  }
  const int val = W.val[li];
  D.pest[i] = (unsigned short)pest;
  D.bit[i] = (unsigned char)((val >> bpn) & 1);
  D.type[i] = sig[0] ? 1 : 0;    // 1 = refinement
  D.s1[i] = (unsigned char)(((pest >> 11) << 1) + (sig[0] ? 1 : 0));
  D.s2[i] = (unsigned char)(32 + (sig[0] ? 1 : 0) + ((sig[1] ? 1 : 0) << 1) + ((sig[2] ? 1 : 0) << 2) + ((sig[3] ? 1 : 0) << 3) +
                            ((sig[4] ? 1 : 0) << 4) + ((sig[5] ? 1 : 0) << 5) + ((sig[6] ? 1 : 0) << 6));
  if (sig[0]) {
    // PredictRef, vle.cpp:81-130
    const int lval = s > 0 ? W.val[li - 1] : 0, lval2 = s > 1 ? W.val[li - 2] : 0;
    const int nval = s < n - 1 ? W.val[li + 1] : 0, nval2 = s < n - 2 ? W.val[li + 2] : 0;
    const int b0 = val >> (bpn + 1), b1 = lval >> bpn, b2 = nval >> (bpn + 1), b3 = lval2 >> bpn, b4 = nval2 >> (bpn + 1);
    const int c0 = (b0 << 1) < b1, c1 = b0 < b2, c2 = (b0 << 1) < b3, c3 = b0 < b4;
    const int x0 = b0 << 1, x1 = b1, x2 = b2 << 1, x3 = b3, x4 = b4 << 1;
    const int xm = (x0 + x1 + x2 + x3 + x4) / 5;
    const int d0 = x0 > xm, d1 = x1 > xm;
    const int ctx1 = (b0 & 15) + ((b1 & 15) << 4) + ((b2 & 15) << 8);
    const int ctx2 = (c0 + (c1 << 1) + (c2 << 2) + (c3 << 3)) + (d0 << 4) + (d1 << 5);
    const int ctx3 = sig[1] + sig[2] + sig[3] + sig[4] + sig[5] + sig[6] + sig[7] + sig[8];
    D.i1[i] = (unsigned short)sig[0]; D.i2[i] = (unsigned short)(ctx1 & 255); D.i3[i] = (unsigned short)ctx2; D.i4[i] = (unsigned short)ctx3;
    D.mix[i] = (unsigned char)(((((pest >> 12) << 1) + d0) << 1) + (b0 & 1));
  } else {
    // PredictSig + CountSig, vle.cpp:144-177
    int ctx1 = 0;
    for (int q = 0; q < 16; q++) if (sig[q + 1]) ctx1 += 1 << q;
    int n1 = 0, n2 = 0;
    for (int d = 1; d <= 32; d++) {
      if (s - d >= 0) { const int m = msb_seen(W, li - d, true, bpn); if (m) n1++; if (m > bpn) n2++; }
      if (s + d < n - 1) { const int m = msb_seen(W, li + d, false, bpn); if (m) n1++; if (m > bpn) n2++; }
    }
    // state&15: the previous four samples of this plane, 1 = coded on the significance path
    int st = 0;
    for (int d = 1; d <= 4; d++) if (s - d >= 0 && !(W.msb[li - d] > bpn)) st |= 1 << (d - 1);
    D.i1[i] = (unsigned short)ctx1; D.i2[i] = (unsigned short)n2; D.i3[i] = 0; D.i4[i] = 0;
    D.mix[i] = (unsigned char)((st << 3) + ((n1 >= 3 ? 3 : n1) << 1) + (n2 > 0 ? 1 : 0));
  }
}

// ---- the adaptive chain for one decision (uniform across the wave)
SA_HD void coder_step(CoderModel &M, CntL *csig0, const short *fwd, const unsigned short *inv, const CoderDesc &D, int i,
                      int bpn, RangeEnc &rc, bool writer) {
  const int bit = D.bit[i], pest = D.pest[i];
  CntL &pl = M.p_laplace[bpn];
  int p1;            // mixer output
  int st[5];
  if (D.type[i]) {
    CntL &c1 = M.cref0[D.i1[i]], &c2 = M.cref1[D.i2[i]], &c3 = M.cref2[D.i3[i]], &c4 = M.cref3[D.i4[i]];
    int *w = M.lmixref[D.mix[i]];
    st[0] = fwd[pest]; st[1] = fwd[pl.p1]; st[2] = fwd[c1.p1]; st[3] = fwd[c2.p1]; st[4] = fwd[c3.p1];
    p1 = mix_predict<5>(w, st, inv);
    // SSE + final mix
    const int sp1 = fwd[p1];
    int q1, q2;
    unsigned short *m1 = M.sse[D.s1[i]][M.sse_lb[D.s1[i]]], *m2 = M.sse[D.s2[i]][M.sse_lb[D.s2[i]]];
    const int pr1 = sse_predict<15>(m1, sp1, &q1);
    const int pr2 = sse_predict<15>(m2, fwd[pr1], &q2);
    int sf[2] = {fwd[(pr1 + pr2 + 1) >> 1], sp1};
    const int p = mix_predict<2>(M.ssemix, sf, inv);
    rc.encode((unsigned)p, bit);
    if (writer) {
      cntl_update(pl, bit, 150); cntl_update(c1, bit, 150); cntl_update(c2, bit, 150); cntl_update(c3, bit, 150); cntl_update(c4, bit, 150);
      mix_update<5>(w, st, p1, bit, 800);
      cnt16_update(m1[q1], bit, 250); cnt16_update(m1[q1 + 1], bit, 250); M.sse_lb[D.s1[i]] = (unsigned char)bit;
      // note: when s1 == s2 cannot happen (s1 < 32 <= s2)
      unsigned short *m2b = M.sse[D.s2[i]][0] + 0;   // re-derive after lb of s1 changed (distinct ctx, so unaffected)
      (void)m2b;
      cnt16_update(m2[q2], bit, 250); cnt16_update(m2[q2 + 1], bit, 250); M.sse_lb[D.s2[i]] = (unsigned char)bit;
      mix_update<2>(M.ssemix, sf, p, bit, 250);
    }
  } else {
    CntL &c1 = csig0[D.i1[i]];
    CntL &c2 = M.csig1[D.i2[i]];
    int *w = M.lmixsig[D.mix[i]];
    st[0] = fwd[pl.p1]; st[1] = fwd[c1.p1]; st[2] = fwd[c2.p1];
    p1 = mix_predict<3>(w, st, inv);
    const int sp1 = fwd[p1];
    int q1, q2;
    unsigned short *m1 = M.sse[D.s1[i]][M.sse_lb[D.s1[i]]], *m2 = M.sse[D.s2[i]][M.sse_lb[D.s2[i]]];
    const int pr1 = sse_predict<15>(m1, sp1, &q1);
    const int pr2 = sse_predict<15>(m2, fwd[pr1], &q2);
    int sf[2] = {fwd[(pr1 + pr2 + 1) >> 1], sp1};
    const int p = mix_predict<2>(M.ssemix, sf, inv);
    rc.encode((unsigned)p, bit);
    if (writer) {
      cntl_update(pl, bit, 150); cntl_update(c1, bit, 300); cntl_update(c2, bit, 300);
      mix_update<3>(w, st, p1, bit, 700);
      cnt16_update(m1[q1], bit, 250); cnt16_update(m1[q1 + 1], bit, 250); M.sse_lb[D.s1[i]] = (unsigned char)bit;
      cnt16_update(m2[q2], bit, 250); cnt16_update(m2[q2 + 1], bit, 250); M.sse_lb[D.s2[i]] = (unsigned char)bit;
      mix_update<2>(M.ssemix, sf, p, bit, 250);
    }
  }
}

// ---- MapEncoder (map.cpp:3-101): 2 x 32768 used-flags, serial
struct MapModel {
  unsigned short cnt[24], cctx[256];
  int mixl[4][5], mixh[4][5], finalmix[2];
  unsigned short sse[2][33];
  int lb;
};
SA_HD void map_model_init(MapModel &m, const unsigned short *inv) {
  for (int i = 0; i < 24; i++) m.cnt[i] = kPScale >> 1;
  for (int i = 0; i < 256; i++) m.cctx[i] = kPScale >> 1;
  for (int a = 0; a < 4; a++) for (int b = 0; b < 5; b++) { m.mixl[a][b] = 0; m.mixh[a][b] = 0; }
  m.finalmix[0] = m.finalmix[1] = 0; m.lb = 0;
  for (int i = 0; i <= 32; i++) { const int x = squash(inv, i * 171 - 2662); m.sse[0][i] = (unsigned short)x; m.sse[1][i] = (unsigned short)x; }   // SSENL<32>: xscale 171
}
SA_HD void map_encode(MapModel &m, const unsigned char *ul, const unsigned char *uh, const short *fwd, const unsigned short *inv, RangeEnc &rc, bool writer) {
  for (int i = 1; i <= 1 << 15; i++) {
    for (int hi = 0; hi < 2; hi++) {
      const unsigned char *a = hi ? uh : ul;
      const int ctx1 = a[i - 1];
      const int ctx2 = hi ? ul[i] : uh[i - 1];
      const int ctx3 = i > 1 ? a[i - 2] : 0;
      const int base = hi ? 12 : 0;
      unsigned short *pc1 = &m.cnt[base + ctx1], *pc2 = &m.cnt[base + 2 + ctx2], *pc3 = &m.cnt[base + 4 + (ctx1 << 1) + ctx3], *pc4 = &m.cnt[base + 8 + (ctx1 << 1) + ctx2];
      int sctx = a[i - 1];
      if (i > 1) sctx += (a[i - 2] << 1);
      if (i > 2) sctx += (a[i - 3] << 2);
      if (i > 3) sctx += (a[i - 4] << 3);
      unsigned short *px = &m.cctx[(hi ? 32 : 0) + sctx];
      int *w = hi ? m.mixh[ctx1 + (ctx3 << 1)] : m.mixl[ctx1 + (ctx3 << 1)];
      int st[5] = {fwd[*pc1], fwd[*pc2], fwd[*pc3], fwd[*pc4], fwd[*px]};
      const int p1 = mix_predict<5>(w, st, inv);
      int q;
      const int sp1 = fwd[p1];
      unsigned short *mp = m.sse[m.lb];
      const int ps = sse_predict<32>(mp, sp1, &q);
      int sf[2] = {fwd[ps], sp1};
      const int p = mix_predict<2>(m.finalmix, sf, inv);
      const int bit = a[i];
      rc.encode((unsigned)p, bit);
      if (writer) {
        cnt16_update(*pc1, bit, 500); cnt16_update(*pc2, bit, 500); cnt16_update(*pc3, bit, 500); cnt16_update(*pc4, bit, 500); cnt16_update(*px, bit, 500);
        mix_update<5>(w, st, p1, bit, 1000);
        cnt16_update(mp[q], bit, 300); cnt16_update(mp[q + 1], bit, 300); m.lb = bit;
        mix_update<2>(m.finalmix, sf, p, bit, 500);
      }
    }
  }
}

// ---- whole stream.  E::nl == 64.  LDS objects are passed in by the caller.
template <class E>
SA_HD int coder_stream(E &ex, const int *s2u, int n, int maxbpn, const unsigned char *used /*nullable: usedl, usedh*/,
                       const unsigned short *laplace, const short *g_fwd, const unsigned short *g_inv, const unsigned short *plap_init,
                       CntL *csig0, unsigned char *out, int cap,
                       CoderModel &M, CoderDesc &D, CoderWin &W, MapModel &MM, short *fwd, unsigned short *inv) {
  // tables -> LDS, model init
  ex.par([&](int l) {
    for (int i = l; i < kPScale; i += E::nl) fwd[i] = g_fwd[i];
    for (int i = l; i < 4095; i += E::nl) inv[i] = g_inv[i];
    for (int i = l; i < 65536; i += E::nl) { csig0[i].p1 = kPScale >> 1; csig0[i].cnt = 0; }
  });
  ex.sync();
  ex.par([&](int l) { coder_model_init(M, inv, plap_init, l, E::nl); });
  ex.sync();
  // The adaptive chain is strictly serial: lane 0 runs it (single-lane LDS writes, no bank
  // conflicts), the other lanes only take part in the parallel context computation.
  RangeEnc rc;
  rc.init(out, cap, true);
  if (used) {
    ex.par([&](int l) {
      if (l == 0) { map_model_init(MM, inv); map_encode(MM, used, used + 32769, fwd, inv, rc, true); }
    });
    ex.sync();
  }
  for (int bpn = maxbpn; bpn >= 0; bpn--) {
    for (int s0 = 0; s0 < n; s0 += kCoderChunk) {
      // stage window [s0-32, s0+64+32)
      ex.par([&](int l) {
        for (int q = l; q < kCoderChunk + 2 * kCoderHalo; q += E::nl) {
          const int k = s0 - kCoderHalo + q;
          const int v = (k >= 0 && k < n) ? s2u[k] : 0;
          W.val[q] = v;
          W.msb[q] = (unsigned char)(v > 0 ? ilog2i(v) : 0);
        }
      });
      ex.sync();
      ex.par([&](int l) { if (s0 + l < n) coder_describe(W, D, l, s0 + l, n, bpn, laplace); });
      ex.sync();
      const int cnt = (n - s0 < kCoderChunk) ? n - s0 : kCoderChunk;
      ex.par([&](int l) {
        if (l == 0)
          for (int i = 0; i < cnt; i++) coder_step(M, csig0, fwd, inv, D, i, bpn, rc, true);
      });
      ex.sync();
    }
  }
  int len = 0;
  ex.par([&](int l) { if (l == 0) { rc.stop(); len = rc.pos; } });
  return len;   // valid on lane 0
}

}  // namespace sacamd
