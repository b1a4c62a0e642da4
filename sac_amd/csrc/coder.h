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
#include <type_traits>
#include "simt.h"
#include "libm_port.h"

namespace sacamd {

constexpr int kPBits = 15, kPScale = 1 << 15, kPScaleM = kPScale - 1;
constexpr int kLaplaceAvg = 1 << 17;      // avg_sum domain of the host table
constexpr int kLaplacePlanes = 18;
constexpr int kCoderChunk = 64;
constexpr int kCoderHalo = 32;
constexpr int kCoderStage = 512;          // range coder output bytes staged in LDS between two flushes

struct CntL { unsigned short p1, cnt; };     // LinearCounterLimit as one 32-bit word

struct CoderModel {
  CntL csig1[80], cref0[32], cref1[256], cref2[64], cref3[160], p_laplace[32];
  int lmixref[32][5], lmixsig[128][3], ssemix[2];
  unsigned short sse[160][2][16];
  unsigned char sse_lb[160];
};

// read-only tables staged in LDS
struct CoderTabs {
  short fwdh[kPScale / 2 + 1]; // LogDomain::Fwd for p <= 16384; Fwd(p) = -Fwd(32768-p) above (odd symmetry)
  unsigned pinv[4097];         // x in [-2048,2048] -> p = clamp(Inv(x)) | Fwd(p) << 16
  unsigned short divt[304];    // PSCALE / (cnt + 3)   (counter.h:40-51)
};

struct CoderWin {            // staged data window of one chunk (with halo)
  int val[kCoderChunk + 2 * kCoderHalo];
  unsigned char msb[kCoderChunk + 2 * kCoderHalo];
  // range coder output of the serial section, copied to the stream by all lanes afterwards
  unsigned char stage[kCoderStage];
  int fl[2];                  // bytes staged, their offset in the output stream
  // wave-parallel decision (coder_step_wave): lanes that own no adaptive element load / store here
  unsigned sinkc[64];
  int sinkw[64];
};

// per-sample descriptor, packed into four ints held by the lane that computed it
struct CoderDescR {
  int a;   // pest | i1 << 16
  int b;   // i2 | i3 << 16
  int c;   // i4 | mix << 16 | s1 << 24
  int d;   // s2 | type << 8 | bit << 9 | fwd(pest) << 16
};

struct RangeEnc {            // RangeCoderSH, encode side
  // Output bytes go to an LDS staging buffer, not to memory: in the serial chain every global store
  // would sit in the same in-order vmcnt queue as the prefetched context word, and waiting for that
  // word would wait for the store acknowledgements of the decision before it.
  unsigned range, FFNum, Cache;
  unsigned long long lowc;
  unsigned char *out, *stage;
  int pos, cap, spos;
  SA_HD void init(unsigned char *o, int capacity, unsigned char *stage_buf) {
    range = 0xFFFFFFFFu; FFNum = 0; Cache = 0; lowc = 0; out = o; pos = 0; cap = capacity; stage = stage_buf; spos = 0;
  }
  SA_HD void flush_serial() {                  // staging buffer full (long carry runs, the map header)
#pragma nounroll
    for (int i = 0; i < spos; i++) if (pos + i < cap) out[pos + i] = stage[i];
    pos += spos; spos = 0;
  }
  SA_HD void put(unsigned b) { stage[spos++] = (unsigned char)b; if (spos == kCoderStage) flush_serial(); }
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

struct RangeDec {            // RangeCoderSH, decode side (model/range.cpp:54-83)
  unsigned range, code;
  const unsigned char *in;   // staged input window (LDS) or the stream itself
  int pos, len;              // bytes consumed / available
  SA_HD unsigned get() { const unsigned b = pos < len ? in[pos] : 0u; pos++; return b; }   // BufIO::GetByte past the end reads the buffer's zero fill
  SA_HD void init(const unsigned char *src, int n) {
    range = 0xFFFFFFFFu; code = 0; in = src; pos = 0; len = n;
    for (int i = 0; i < 5; i++) code = (code << 8) + get();      // NUM + 1 bytes; the first is the encoder's initial Cache (0)
  }
  SA_HD int decode(unsigned p1) {
    const unsigned rnew = (unsigned)(((unsigned long long)range * ((unsigned)(kPScale - p1) << (32 - kPBits))) >> 32);
    const int bit = code >= rnew;
    if (bit) { range -= rnew; code -= rnew; } else range = rnew;
    while (range < 0x01000000u) { range <<= 8; code = (code << 8) + get(); }
    return bit;
  }
};

// round-half-away-from-zero shifts (val < 0 ? -((-val + h) >> s) : (val + h) >> s), written without a
// branch: the lanes of a wave-parallel decision differ in sign
SA_HD int idiv_s(int val, int s) { const int m = val >> 31; const int r = (((val ^ m) - m) + (1 << (s - 1))) >> s; return (r ^ m) - m; }
SA_HD int idiv_s64(long long val, int s) { const long long m = val >> 63; const long long r = (((val ^ m) - m) + (1LL << (s - 1))) >> s; return (int)((r ^ m) - m); }
SA_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
SA_HD int ilog2i(int v) { int nb = 0; while (v >>= 1) nb++; return nb; }

// LinearCounterLimit::update (counter.h:58-68) on a loaded value; returns the new packed word
SA_HD CntL cntl_next(CntL c, int bit, int limit, const unsigned short *divt) {
  unsigned cnt = c.cnt;
  if ((int)cnt < limit) cnt++;
  const int d = divt[cnt];
  const int p1 = c.p1;
  const int dp = bit ? ((kPScale - p1) * d) >> kPBits : -((p1 * d) >> kPBits);
  CntL r; r.p1 = (unsigned short)clampi(p1 + dp, 1, kPScaleM); r.cnt = (unsigned short)cnt;
  return r;
}
SA_HD unsigned short cnt16_next(int p1, int bit, int L) {       // LinearCounter16::update(bit,L), counter.h:31-37
  const int err = (bit << kPBits) - p1;
  return (unsigned short)clampi(p1 + idiv_s(L * err, kPBits), 1, kPScaleM);
}
SA_HD void cnt16_update(unsigned short &p1, int bit, int L) { p1 = cnt16_next(p1, bit, L); }
// squash+stretch: x -> packed (p | Fwd(p) << 16)
SA_HD unsigned pinv_lookup(const unsigned *pinv, int x) { return pinv[clampi(x, -2048, 2048) + 2048]; }
SA_HD int squash(const unsigned *pinv, int x) { return (int)(pinv_lookup(pinv, x) & 0xffff); }

template <int N>
SA_HD int mix_dot(const int *w, const int *st) {                 // mixer.h:76-85 (returns the stretched-domain sum)
  long long sum = 0;
#pragma unroll
  for (int i = 0; i < N; i++) sum += (long long)(w[i] * st[i]);
  return idiv_s64(sum, 16);
}
template <int N>
SA_HD void mix_next(int *w, const int *st, int pd, int bit, int rate) {   // mixer.h:88-96, in place on loaded weights
  const int err = (bit << kPBits) - pd;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int de = idiv_s(st[i] * err, 12);
    const int wd = idiv_s(de * rate, 12);
    w[i] = clampi(w[i] + wd, -(1 << 19), (1 << 19) - 1);
  }
}
// SSENL<N>::Predict (sse.h:101-113) split: bin index / interpolation
template <int N> SA_HD void sse_bin(int stp, int *pq, int *pmod) {
  constexpr int tscale = 2662, xscale = (2 * tscale) / (N - 1);
  int q = stp + tscale;
  q = q < 0 ? 0 : (q > 2 * tscale ? 2 * tscale : q);
  *pq = q / xscale; *pmod = q - (*pq) * xscale;
}
template <int N> SA_HD int sse_interp(int pl, int ph, int pmod) {
  constexpr int xscale = (2 * 2662) / (N - 1);
  return clampi((pl * (xscale - pmod) + ph * pmod) / xscale, 1, kPScaleM);
}

SA_HD int fwd_at(const CoderTabs &T, int p) {      // one load, no branch (lanes of a wave-parallel decision differ in p)
  const bool lo = p <= kPScale / 2;
  const int v = T.fwdh[lo ? p : kPScale - p];
  return lo ? v : -v;
}

SA_HD void coder_tabs_init(CoderTabs &T, const short *g_fwd, const unsigned short *g_inv, int lane, int nl) {
  for (int i = lane; i <= kPScale / 2; i += nl) T.fwdh[i] = g_fwd[i];
  for (int i = lane; i < 4097; i += nl) {
    const int x = i - 2048;
    const int p = x < -2047 ? 1 : (x > 2047 ? kPScaleM : clampi((int)g_inv[x + 2047], 1, kPScaleM));
    T.pinv[i] = (unsigned)p | ((unsigned)(unsigned short)g_fwd[p] << 16);   // full-range source table
  }
  for (int i = lane; i < 304; i += nl) T.divt[i] = (unsigned short)(kPScale / (i + 3));
}

SA_HD void coder_model_init(CoderModel &m, const CoderTabs &T, const unsigned short *plap_init, int lane, int nl) {
  auto fill = [&](CntL *a, int n) { for (int i = lane; i < n; i += nl) { a[i].p1 = kPScale >> 1; a[i].cnt = 0; } };
  fill(m.csig1, 80); fill(m.cref0, 32); fill(m.cref1, 256); fill(m.cref2, 64); fill(m.cref3, 160);
  for (int i = lane; i < 32; i += nl) { m.p_laplace[i].p1 = plap_init[i]; m.p_laplace[i].cnt = 0; }
  for (int i = lane; i < 32 * 5; i += nl) (&m.lmixref[0][0])[i] = 0;
  for (int i = lane; i < 128 * 3; i += nl) (&m.lmixsig[0][0])[i] = 0;
  if (lane < 2) m.ssemix[lane] = 0;
  for (int i = lane; i < 160 * 2 * 16; i += nl) {
    const int k = i & 15;
    (&m.sse[0][0][0])[i] = (unsigned short)squash(T.pinv, k * 380 - 2662);   // SSENL<15>: xscale 380, tscale 2662 (sse.h:91-99)
  }
  for (int i = lane; i < 160; i += nl) m.sse_lb[i] = 0;
}

// ---- per-sample context computation (data only).  W is the staged window of the chunk that
// starts at sample s0; local index of sample s is s - s0 + kCoderHalo.
SA_HD int msb_seen(const CoderWin &W, int li, bool before, int bpn) {
  const int m = W.msb[li];
  return before ? (m >= bpn ? m : 0) : (m > bpn ? m : 0);
}

// BitplaneCoder::PredictLaplace (vle.cpp:70-79) evaluated directly, for the (avg_sum, plane) pairs the host table does not
// hold: avg_sum >= 2^17 or plane >= 18, i.e. material wider than 16 bits.  exp / pow are the glibc ports of libm_port.h, so
// the value is the one the table would hold.
SA_HD int laplace_direct(unsigned avg, int bpn) {
  double p_l = 0.0;
  if (avg > 0) {
    const double theta = sa_exp(-1.0 / (double)avg);
    p_l = 1.0 - 1.0 / (1 + sa_pow(theta, (double)(1 << bpn)));
  }
  return clampi((int)round(p_l * kPScale), 1, kPScaleM);
}

SA_HD CoderDescR coder_describe(const CoderWin &W, const CoderTabs &T, int i, int s, int n, int bpn, const unsigned short *laplace) {
  const int li = i + kCoderHalo;
  // GetAvgSum(32), vle.cpp:54-68
  unsigned long long nsum = 0; int nidx = 0;
  const unsigned ml = ~((1u << bpn) - 1), mr = ~((1u << (bpn + 1)) - 1);
  for (int d = -32; d <= 32; d++) {
    const int k = s + d;
    if (k >= 0 && k < n) { nsum += (unsigned)W.val[li + d] & (d < 0 ? ml : mr); nidx++; }
  }
  const unsigned avg = nidx > 0 ? (unsigned)((nsum + (nidx - 1)) / nidx) : 0;
  const int pest = (avg < (unsigned)kLaplaceAvg && bpn < kLaplacePlanes) ? laplace[(size_t)bpn * kLaplaceAvg + avg] : laplace_direct(avg, bpn);
  // GetSigState, vle.cpp:33-52
  int sig[17];
  sig[0] = msb_seen(W, li, false, bpn);
  for (int d = 1; d <= 8; d++) {
    sig[2 * d - 1] = (s > d - 1) ? msb_seen(W, li - d, true, bpn) : 0;
    sig[2 * d] = (s < n - d) ? msb_seen(W, li + d, false, bpn) : 0;
  }
  const int val = W.val[li];
  const int bit = (val >> bpn) & 1;
  const int s1 = ((pest >> 11) << 1) + (sig[0] ? 1 : 0);
  const int s2 = 32 + (sig[0] ? 1 : 0) + ((sig[1] ? 1 : 0) << 1) + ((sig[2] ? 1 : 0) << 2) + ((sig[3] ? 1 : 0) << 3) +
                 ((sig[4] ? 1 : 0) << 4) + ((sig[5] ? 1 : 0) << 5) + ((sig[6] ? 1 : 0) << 6);
  int i1, i2, i3 = 0, i4 = 0, mix, type;
  if (sig[0]) {
    // PredictRef, vle.cpp:81-130
    type = 1;
    const int lval = s > 0 ? W.val[li - 1] : 0, lval2 = s > 1 ? W.val[li - 2] : 0;
    const int nval = s < n - 1 ? W.val[li + 1] : 0, nval2 = s < n - 2 ? W.val[li + 2] : 0;
    const int b0 = val >> (bpn + 1), b1 = lval >> bpn, b2 = nval >> (bpn + 1), b3 = lval2 >> bpn, b4 = nval2 >> (bpn + 1);
    const int c0 = (b0 << 1) < b1, c1 = b0 < b2, c2 = (b0 << 1) < b3, c3 = b0 < b4;
    const int x0 = b0 << 1, x1 = b1, x2 = b2 << 1, x3 = b3, x4 = b4 << 1;
    const int xm = (x0 + x1 + x2 + x3 + x4) / 5;
    const int d0 = x0 > xm, d1 = x1 > xm;
    const int ctx1 = (b0 & 15) + ((b1 & 15) << 4) + ((b2 & 15) << 8);
    i1 = sig[0]; i2 = ctx1 & 255;
    i3 = (c0 + (c1 << 1) + (c2 << 2) + (c3 << 3)) + (d0 << 4) + (d1 << 5);
    i4 = sig[1] + sig[2] + sig[3] + sig[4] + sig[5] + sig[6] + sig[7] + sig[8];
    if (i4 > 159) i4 = 159;   // cref3 is updated but never predicts (vle.cpp:122,127,138): any slot does; 16-bit material stays below 137
    mix = ((((pest >> 12) << 1) + d0) << 1) + (b0 & 1);
  } else {
    // PredictSig + CountSig, vle.cpp:144-177
    type = 0;
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
    i1 = ctx1; i2 = n2;
    mix = (st << 3) + ((n1 >= 3 ? 3 : n1) << 1) + (n2 > 0 ? 1 : 0);
  }
  CoderDescR D;
  D.a = pest | (i1 << 16);
  D.b = i2 | (i3 << 16);
  D.c = i4 | (mix << 16) | (s1 << 24);
  D.d = s2 | (type << 8) | (bit << 9) | ((int)(unsigned short)fwd_at(T, pest) << 16);
  return D;
}

// ---- the adaptive chain for one decision.  c1sig: the (possibly prefetched) csig0 entry for a
// significance decision; returns its updated value through *c1out.
// RC = RangeEnc: the bit is D's; RC = RangeDec: the bit comes from the stream (BitplaneCoder::Decode, vle.cpp:233-261).  Returns it.
template <class RC>
SA_HD int coder_step(CoderModel &M, const CoderTabs &T, CoderDescR D, int bpn, CntL c1sig, CntL *c1out, RC &rc) {
  const int pest = D.a & 0xffff, i1 = (D.a >> 16) & 0xffff, i2 = D.b & 0xffff, i3 = (D.b >> 16) & 0xffff;
  const int i4 = D.c & 0xffff, mixc = (D.c >> 16) & 0xff, s1 = (D.c >> 24) & 0xff, s2 = D.d & 0xff;
  const int type = (D.d >> 8) & 1, st_pest = (short)(D.d >> 16);
  int bit = (D.d >> 9) & 1;
  (void)pest;
  // ---- round 1: every state word this decision touches
  const CntL pl = M.p_laplace[bpn];
  const int lb1 = M.sse_lb[s1], lb2 = M.sse_lb[s2];
  int sw[2] = {M.ssemix[0], M.ssemix[1]};
  unsigned short *m1 = M.sse[s1][lb1], *m2 = M.sse[s2][lb2];
  int st[5], w[5];
  CntL c1, c2, c3 = pl, c4 = pl;
  int x;
  if (type) {
    c1 = M.cref0[i1]; c2 = M.cref1[i2]; c3 = M.cref2[i3]; c4 = M.cref3[i4];
    const int *wp = M.lmixref[mixc];
#pragma unroll
    for (int q = 0; q < 5; q++) w[q] = wp[q];
    st[0] = st_pest; st[1] = fwd_at(T, pl.p1); st[2] = fwd_at(T, c1.p1); st[3] = fwd_at(T, c2.p1); st[4] = fwd_at(T, c3.p1);
    x = mix_dot<5>(w, st);
  } else {
    c1 = c1sig; c2 = M.csig1[i2];
    const int *wp = M.lmixsig[mixc];
#pragma unroll
    for (int q = 0; q < 3; q++) w[q] = wp[q];
    st[0] = fwd_at(T, pl.p1); st[1] = fwd_at(T, c1.p1); st[2] = fwd_at(T, c2.p1); st[3] = 0; st[4] = 0;
    x = mix_dot<3>(w, st);
  }
  const unsigned pk = pinv_lookup(T.pinv, x);
  const int p1 = (int)(pk & 0xffff), sp1 = (short)(pk >> 16);
  int q1, r1, q2, r2;
  sse_bin<15>(sp1, &q1, &r1);
  const int m1a = m1[q1], m1b = m1[q1 + 1];
  const int pr1 = sse_interp<15>(m1a, m1b, r1);
  sse_bin<15>(fwd_at(T, pr1), &q2, &r2);
  const int m2a = m2[q2], m2b = m2[q2 + 1];
  const int pr2 = sse_interp<15>(m2a, m2b, r2);
  int sf[2] = {fwd_at(T, (pr1 + pr2 + 1) >> 1), sp1};
  const int p = (int)(pinv_lookup(T.pinv, mix_dot<2>(sw, sf)) & 0xffff);
  if constexpr (std::is_same<RC, RangeDec>::value) bit = rc.decode((unsigned)p); else rc.encode((unsigned)p, bit);
  // ---- updates (computed from the loaded values; stores only)
  M.p_laplace[bpn] = cntl_next(pl, bit, 150, T.divt);
  if (type) {
    M.cref0[i1] = cntl_next(c1, bit, 150, T.divt); M.cref1[i2] = cntl_next(c2, bit, 150, T.divt);
    M.cref2[i3] = cntl_next(c3, bit, 150, T.divt); M.cref3[i4] = cntl_next(c4, bit, 150, T.divt);
    mix_next<5>(w, st, p1, bit, 800);
    int *wp = M.lmixref[mixc];
#pragma unroll
    for (int q = 0; q < 5; q++) wp[q] = w[q];
  } else {
    *c1out = cntl_next(c1, bit, 300, T.divt);
    M.csig1[i2] = cntl_next(c2, bit, 300, T.divt);
    mix_next<3>(w, st, p1, bit, 700);
    int *wp = M.lmixsig[mixc];
#pragma unroll
    for (int q = 0; q < 3; q++) wp[q] = w[q];
  }
  m1[q1] = cnt16_next(m1a, bit, 250); m1[q1 + 1] = cnt16_next(m1b, bit, 250); M.sse_lb[s1] = (unsigned char)bit;
  m2[q2] = cnt16_next(m2a, bit, 250); m2[q2 + 1] = cnt16_next(m2b, bit, 250); M.sse_lb[s2] = (unsigned char)bit;
  mix_next<2>(sw, sf, p, bit, 250);
  M.ssemix[0] = sw[0]; M.ssemix[1] = sw[1];
  return bit;
}

#if defined(__HIPCC__)
// ---- the same decision spread over the lanes of the wave (device only).  A lone lane issues one
// instruction every four cycles or more whatever it computes, so the cost of a decision is its
// instruction count: here every adaptive element is owned by a lane -- lane q holds mixer input q
// and its weight (ref: pest, p_laplace, cref0..2 on lanes 0..4, cref3 on lane 5; sig: p_laplace,
// csig0, csig1 on lanes 0..2), even / odd lanes hold the two SSE bins -- the dot product and the two
// interpolations are DPP sums, and everything else is computed redundantly by all lanes.  Integer
// arithmetic throughout, so the regrouping is exact; loads precede stores exactly as in coder_step.
// `cur`: the prefetched csig0 word (all lanes); *upd: its update (uniform).  Only lane 0's rc is live.
template <int LANE> __device__ __forceinline__ int coder_wlane(int sval, int old) {   // old with lane LANE replaced by the uniform sval
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "i"(LANE));
  return old;
}
template <int CTRL> __device__ __forceinline__ int coder_dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ void coder_step_wave(CoderModel &M, const CoderTabs &T, CoderWin &W, CoderDescR D, int bpn, unsigned cur,
                                                unsigned *upd, RangeEnc &rc, int L) {
  const int i1 = (D.a >> 16) & 0xffff, i2 = D.b & 0xffff, i3 = (D.b >> 16) & 0xffff;
  const int i4 = D.c & 0xffff, mixc = (D.c >> 16) & 0xff, s1 = (D.c >> 24) & 0xff, s2 = D.d & 0xff;
  const int type = (D.d >> 8) & 1, bit = (D.d >> 9) & 1, st_pest = (short)(D.d >> 16);
  // element addresses as byte offsets from M; uniform candidates are written into their owner's lane
  char *const mb = reinterpret_cast<char *>(&M);
  auto off = [&](const void *q) { return __builtin_amdgcn_readfirstlane((int)(reinterpret_cast<const char *>(q) - mb)); };
  int oc = (int)(reinterpret_cast<char *>(&W.sinkc[L]) - mb);
  int ow = (int)(reinterpret_cast<char *>(&W.sinkw[L]) - mb);
  if (type) {
    oc = coder_wlane<1>(off(&M.p_laplace[bpn]), oc);
    oc = coder_wlane<2>(off(&M.cref0[i1]), oc);
    oc = coder_wlane<3>(off(&M.cref1[i2]), oc);
    oc = coder_wlane<4>(off(&M.cref2[i3]), oc);
    oc = coder_wlane<5>(off(&M.cref3[i4]), oc);
    const int wb = off(&M.lmixref[mixc][0]) + 4 * L;
    ow = L < 5 ? wb : ow;
  } else {
    oc = coder_wlane<0>(off(&M.p_laplace[bpn]), oc);
    oc = coder_wlane<2>(off(&M.csig1[i2]), oc);
    const int wb = off(&M.lmixsig[mixc][0]) + 4 * L;
    ow = L < 3 ? wb : ow;
  }
  unsigned *cp = reinterpret_cast<unsigned *>(mb + oc);
  int *wq = reinterpret_cast<int *>(mb + ow);
  // ---- loads
  unsigned cw = *cp;
  if (!type && L == 1) cw = cur;
  int w = *wq;
  const int lb1 = M.sse_lb[s1], lb2 = M.sse_lb[s2];
  int sw[2] = {M.ssemix[0], M.ssemix[1]};
  int st = fwd_at(T, (int)(cw & 0xffff));
  if (type && L == 0) st = st_pest;
  if (L >= (type ? 5 : 3)) st = 0;
  // mixer: sum of the products as (high, low 16 bits) pairs, reduced over lanes 0..7 of the row
  const int prod = w * st;
  int sh = prod >> 16, sl = prod & 0xffff;
  sh += coder_dpp0<0x111>(sh); sl += coder_dpp0<0x111>(sl);     // row_shr:1
  sh += coder_dpp0<0x112>(sh); sl += coder_dpp0<0x112>(sl);     // row_shr:2
  sh += coder_dpp0<0x114>(sh); sl += coder_dpp0<0x114>(sl);     // row_shr:4
  const int Sh = __builtin_amdgcn_readlane(sh, 7), Sl = __builtin_amdgcn_readlane(sl, 7);
  // idiv_s64(Sh * 65536 + Sl, 16)
  const int H = Sh + (Sl >> 16), Lo = Sl & 0xffff;
  const int x = H + ((H >= 0) ? (Lo >= 32768 ? 1 : 0) : (Lo > 32768 ? 1 : 0));
  const unsigned pk = pinv_lookup(T.pinv, x);
  const int p1 = (int)(pk & 0xffff), sp1 = (short)(pk >> 16);
  constexpr int xs = (2 * 2662) / 14;
  const int odd = L & 1;
  int q1, r1, q2, r2;
  sse_bin<15>(sp1, &q1, &r1);
  unsigned short *m1 = &M.sse[s1][lb1][q1 + odd];
  const int my1 = *m1;
  int t1 = my1 * (odd ? r1 : xs - r1);
  t1 += coder_dpp0<0xB1>(t1);                                   // quad_perm [1,0,3,2]
  const int pr1 = clampi(t1 / xs, 1, kPScaleM);
  sse_bin<15>(fwd_at(T, pr1), &q2, &r2);
  unsigned short *m2 = &M.sse[s2][lb2][q2 + odd];
  const int my2 = *m2;
  int t2 = my2 * (odd ? r2 : xs - r2);
  t2 += coder_dpp0<0xB1>(t2);
  const int pr2 = clampi(t2 / xs, 1, kPScaleM);
  int sf[2] = {fwd_at(T, (pr1 + pr2 + 1) >> 1), sp1};
  const int p = (int)(pinv_lookup(T.pinv, mix_dot<2>(sw, sf)) & 0xffff);
  // ---- updates: each owner stores its element, the others their sink word
  CntL c; c.p1 = (unsigned short)(cw & 0xffff); c.cnt = (unsigned short)(cw >> 16);
  const CntL cn = cntl_next(c, bit, (type || L == 0) ? 150 : 300, T.divt);   // p_laplace (lane 0 of a sig decision) keeps limit 150
  const unsigned nw = (unsigned)cn.p1 | ((unsigned)cn.cnt << 16);
  *cp = nw;
  *upd = (unsigned)__builtin_amdgcn_readlane((int)nw, 1);
  mix_next<1>(&w, &st, p1, bit, type ? 800 : 700);
  *wq = w;
  unsigned short *h1 = L < 2 ? m1 : reinterpret_cast<unsigned short *>(&W.sinkw[L]);
  unsigned short *h2 = L < 2 ? m2 : reinterpret_cast<unsigned short *>(&W.sinkw[L]);
  *h1 = cnt16_next(my1, bit, 250);
  *h2 = cnt16_next(my2, bit, 250);
  mix_next<2>(sw, sf, p, bit, 250);
  if (L == 0) {
    rc.encode((unsigned)p, bit);
    M.sse_lb[s1] = (unsigned char)bit; M.sse_lb[s2] = (unsigned char)bit;
    M.ssemix[0] = sw[0]; M.ssemix[1] = sw[1];
  }
  asm volatile("" ::: "memory");        // decisions stay in program order (elements change owners between decisions)
}
#endif

// ---- MapEncoder (map.cpp:3-101): 2 x 32768 used-flags, serial
struct MapModel {
  unsigned short cnt[24], cctx[256];
  int mixl[4][5], mixh[4][5], finalmix[2];
  unsigned short sse[2][33];
  int lb;
};
SA_HD void map_model_init(MapModel &m, const unsigned *pinv) {
  for (int i = 0; i < 24; i++) m.cnt[i] = kPScale >> 1;
  for (int i = 0; i < 256; i++) m.cctx[i] = kPScale >> 1;
  for (int a = 0; a < 4; a++) for (int b = 0; b < 5; b++) { m.mixl[a][b] = 0; m.mixh[a][b] = 0; }
  m.finalmix[0] = m.finalmix[1] = 0; m.lb = 0;
  for (int i = 0; i <= 32; i++) { const int x = squash(pinv, i * 171 - 2662); m.sse[0][i] = (unsigned short)x; m.sse[1][i] = (unsigned short)x; }   // SSENL<32>: xscale 171
}
// RC = RangeEnc: MapEncoder::Encode over the flags ul / uh; RC = RangeDec: MapEncoder::Decode (map.cpp:87-101) INTO them
// (ul[0] / uh[0] are never coded and must be 0 on entry)
template <class RC, class U>
SA_HD void map_code(MapModel &m, U *ul, U *uh, const CoderTabs &T, RC &rc) {
  const unsigned *pinv = T.pinv;
  for (int i = 1; i <= 1 << 15; i++) {
    for (int hi = 0; hi < 2; hi++) {
      U *a = hi ? uh : ul;
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
      int st[5] = {fwd_at(T, *pc1), fwd_at(T, *pc2), fwd_at(T, *pc3), fwd_at(T, *pc4), fwd_at(T, *px)};
      const unsigned pk = pinv_lookup(pinv, mix_dot<5>(w, st));
      const int p1 = (int)(pk & 0xffff), sp1 = (short)(pk >> 16);
      int q, r;
      unsigned short *mp = m.sse[m.lb];
      sse_bin<32>(sp1, &q, &r);
      const int ps = sse_interp<32>(mp[q], mp[q + 1], r);
      int sf[2] = {fwd_at(T, ps), sp1};
      const int p = (int)(pinv_lookup(pinv, mix_dot<2>(m.finalmix, sf)) & 0xffff);
      int bit;
      if constexpr (std::is_same<RC, RangeDec>::value) { bit = rc.decode((unsigned)p); a[i] = (unsigned char)bit; }
      else { bit = a[i]; rc.encode((unsigned)p, bit); }
      cnt16_update(*pc1, bit, 500); cnt16_update(*pc2, bit, 500); cnt16_update(*pc3, bit, 500); cnt16_update(*pc4, bit, 500); cnt16_update(*px, bit, 500);
      mix_next<5>(w, st, p1, bit, 1000);
      cnt16_update(mp[q], bit, 300); cnt16_update(mp[q + 1], bit, 300); m.lb = bit;
      mix_next<2>(m.finalmix, sf, p, bit, 500);
    }
  }
}

// ---- whole stream.  E::nl == 64.  LDS objects are passed in by the caller.
template <class E>
SA_HD int coder_stream(E &ex, const int *s2u, int n, int maxbpn, const unsigned char *used /*nullable: usedl, usedh*/,
                       const unsigned short *laplace, const short *g_fwd, const unsigned short *g_inv, const unsigned short *plap_init,
                       CntL *csig0, unsigned char *out, int cap, CoderModel &M, const CoderTabs &T, CoderWin &W, MapModel &MM,
                       int serial_chain = 0 /*device, parity tap: run the decision chain on lane 0 (coder_step) instead of coder_step_wave*/) {
  (void)g_fwd; (void)g_inv; (void)serial_chain;
  // T (read-only tables) has been staged by the caller (coder_tabs_init) and may be shared by
  // several streams of one workgroup
  ex.par([&](int l) {
    for (int i = l; i < 65536; i += E::nl) { csig0[i].p1 = kPScale >> 1; csig0[i].cnt = 0; }
  });
  ex.sync();
  ex.par([&](int l) { coder_model_init(M, T, plap_init, l, E::nl); if (l == 0) { W.fl[0] = 0; W.fl[1] = 0; }
    for (int q = l; q < 64; q += E::nl) { W.sinkc[q] = 0; W.sinkw[q] = 0; } });
  ex.sync();
  // The adaptive chain is strictly serial: lane 0 runs it (single-lane LDS traffic), the other
  // lanes take part in the parallel context computation and keep their sample's descriptor in
  // registers, from where lane 0 fetches it with v_readlane.
  RangeEnc rc;
  rc.init(out, cap, W.stage);
  // hand the bytes staged so far to the next parallel section (lane 0 only)
  auto publish = [&]() { W.fl[0] = rc.spos; W.fl[1] = rc.pos; rc.pos += rc.spos; rc.spos = 0; };
  auto flush_par = [&](int l) {
    const int cntb = W.fl[0], at = W.fl[1];
    for (int q = l; q < cntb; q += E::nl) if (at + q < cap) out[at + q] = W.stage[q];
  };
  if (used) {
    ex.par([&](int l) {
      if (l == 0) { map_model_init(MM, T.pinv); map_code(MM, used, used + 32769, T, rc); publish(); }
    });
    ex.sync();
  }
  typename E::template Reg<int> da, db, dc, dd;
  for (int bpn = maxbpn; bpn >= 0; bpn--) {
    for (int s0 = 0; s0 < n; s0 += kCoderChunk) {
      ex.par([&](int l) {
        flush_par(l);
        for (int q = l; q < kCoderChunk + 2 * kCoderHalo; q += E::nl) {
          const int k = s0 - kCoderHalo + q;
          const int v = (k >= 0 && k < n) ? s2u[k] : 0;
          W.val[q] = v;
          W.msb[q] = (unsigned char)(v > 0 ? ilog2i(v) : 0);
        }
      });
      ex.sync();
      ex.par([&](int l) {
        CoderDescR D{0, 0, 0, 0};
        if (s0 + l < n) D = coder_describe(W, T, l, s0 + l, n, bpn, laplace);
        da[l] = D.a; db[l] = D.b; dc[l] = D.c; dd[l] = D.d;
      });
      const int cnt = (n - s0 < kCoderChunk) ? n - s0 : kCoderChunk;
      // serial chain; the csig0 word of the NEXT significance decision is fetched one decision
      // ahead (HBM/L2 latency) and forwarded from the register when both hit the same context.
      // The store of a decision's csig0 update is issued at the start of the following decision,
      // right before that prefetch: both then have a whole decision to complete, and the prefetch,
      // issued after the store, observes it.
      bool chain_done = false;
#if defined(__HIPCC__)
      if constexpr (E::is_device) if (!serial_chain) {
        chain_done = true;
        ex.par([&](int l) {
          unsigned *cs = reinterpret_cast<unsigned *>(csig0);
          // csig0 index of decision k (-1: a refinement decision), from the lane that described it
          auto sig_index = [&](int k) { const int a = ex.lane_geti(da, k), d = ex.lane_geti(dd, k); return ((d >> 8) & 1) ? -1 : ((a >> 16) & 0xffff); };
          int idx = sig_index(0);
          unsigned cur = cs[idx < 0 ? 0 : idx];
          asm volatile("" : "+v"(cur));            // no load is pending when the loop is entered
          bool pend = false; int pidx = 0; unsigned pval = 0;
          for (int i = 0; i < cnt; i++) {
            if (pend) { if (l == 0) cs[pidx] = pval; pend = false; }
            const CoderDescR D{ex.lane_geti(da, i), ex.lane_geti(db, i), ex.lane_geti(dc, i), ex.lane_geti(dd, i)};
            unsigned nxt = cur; int idxn = -1;
            if (i + 1 < cnt) { idxn = sig_index(i + 1); if (idxn >= 0) nxt = cs[idxn]; }
            unsigned upd = cur;
            coder_step_wave(M, T, W, D, bpn, cur, &upd, rc, l);
            if (idx >= 0) {
              pend = true; pidx = idx; pval = upd;
              if (idxn == idx) nxt = upd;
            }
            asm volatile("" : "+v"(nxt));          // the prefetch completes here, not behind the next decision's store
            cur = nxt; idx = idxn;
          }
          if (l == 0) { if (pend) cs[pidx] = pval; publish(); }
        });
      }
#endif
      if (!chain_done) {
        ex.lane0([&]() {
          CoderDescR D{ex.lane_geti(da, 0), ex.lane_geti(db, 0), ex.lane_geti(dc, 0), ex.lane_geti(dd, 0)};
          int idx = (D.a >> 16) & 0xffff;
          CntL cur = csig0[((D.d >> 8) & 1) ? 0 : idx];
          bool pend = false; int pidx = 0; CntL pval = cur;
          for (int i = 0; i < cnt; i++) {
            if (pend) { csig0[pidx] = pval; pend = false; }
            CoderDescR Dn = D; CntL nxt = cur; int idxn = idx;
            const bool has_next = i + 1 < cnt;
            if (has_next) {
              Dn = CoderDescR{ex.lane_geti(da, i + 1), ex.lane_geti(db, i + 1), ex.lane_geti(dc, i + 1), ex.lane_geti(dd, i + 1)};
              idxn = (Dn.a >> 16) & 0xffff;
              if (!((Dn.d >> 8) & 1)) nxt = csig0[idxn];
            }
            const bool is_sig = !((D.d >> 8) & 1);
            CntL upd = cur;
            coder_step(M, T, D, bpn, cur, &upd, rc);
            if (is_sig) {
              pend = true; pidx = idx; pval = upd;
              if (has_next && !((Dn.d >> 8) & 1) && idxn == idx) nxt = upd;
            }
            D = Dn; cur = nxt; idx = idxn;
          }
          if (pend) csig0[pidx] = pval;
          publish();
        });
      }
      ex.sync();
    }
  }
  ex.par([&](int l) { flush_par(l); });
  ex.sync();
  int len = 0;
  ex.lane0([&]() { rc.stop(); rc.flush_serial(); len = rc.pos; });
  return len;   // valid on lane 0
}

// ---- whole stream, decode side: BitplaneCoder::Decode (vle.cpp:233-261) + RangeCoderSH::DecodeBitOne [+ MapEncoder::Decode].
// E::nl == 64.  The contexts of a sample in plane b depend on the bits of ITS LEFT NEIGHBOURS in the same plane, so -- unlike
// the encoder -- nothing can be described ahead: every decision describes its sample from the window of PARTIAL values
// (bits above the plane for the sample and its right neighbours, bits down to the plane for its left neighbours; masking
// a partial value with the encoder's masks leaves it unchanged, so coder_describe is used as it is) and then runs the
// adaptive chain; lane 0 does both.  The window of a 64-sample chunk (+ 32 either side) is staged in LDS, updated in place
// and written back to `s2u` (global, zero-initialised here) at the end of the chunk; the input bytes of a chunk are staged
// in LDS too (a chunk consumes at most 2 bytes per decision).  Returns the number of input bytes consumed (lane 0).
constexpr int kDecStage = 256;
template <class E>
SA_HD int coder_stream_dec(E &ex, const unsigned char *in, int inlen, int n, int maxbpn, unsigned char *used_out /*nullable: usedl, usedh [2][32769]*/,
                           const unsigned short *laplace, const unsigned short *plap_init, CntL *csig0, int *s2u /*out [n]*/,
                           CoderModel &M, const CoderTabs &T, CoderWin &W, MapModel &MM) {
  static_assert(kDecStage <= kCoderStage, "input staging shares the encoder's output staging buffer");
  ex.par([&](int l) {
    for (int i = l; i < 65536; i += E::nl) { csig0[i].p1 = kPScale >> 1; csig0[i].cnt = 0; }
    for (int i = l; i < n; i += E::nl) s2u[i] = 0;
    if (used_out) for (int i = l; i < 2 * 32769; i += E::nl) used_out[i] = 0;
  });
  ex.gsync();
  ex.par([&](int l) { coder_model_init(M, T, plap_init, l, E::nl); if (l == 0) { W.fl[0] = 0; W.fl[1] = 0; } });
  ex.sync();
  RangeDec rc;
  rc.init(in, inlen);                       // the five priming bytes and the map header come straight from the stream
  if (used_out) {
    ex.lane0([&]() { map_model_init(MM, T.pinv); map_code(MM, used_out, used_out + 32769, T, rc); });
    ex.gsync();
  }
  int consumed = rc.pos;                    // stream position of W.stage[0] (uniform: broadcast from lane 0 through LDS)
  ex.lane0([&]() { W.fl[0] = rc.pos; });
  ex.sync();
  consumed = W.fl[0];
  for (int bpn = maxbpn; bpn >= 0; bpn--) {
    for (int s0 = 0; s0 < n; s0 += kCoderChunk) {
      ex.par([&](int l) {
        for (int q = l; q < kCoderChunk + 2 * kCoderHalo; q += E::nl) {
          const int k = s0 - kCoderHalo + q;
          const int v = (k >= 0 && k < n) ? s2u[k] : 0;
          W.val[q] = v;
          W.msb[q] = (unsigned char)(v > 0 ? ilog2i(v) : 0);
        }
        for (int q = l; q < kDecStage; q += E::nl) W.stage[q] = consumed + q < inlen ? in[consumed + q] : (unsigned char)0;
      });
      ex.sync();
      const int cnt = (n - s0 < kCoderChunk) ? n - s0 : kCoderChunk;
      ex.lane0([&]() {
        rc.in = W.stage; rc.pos = 0; rc.len = kDecStage;
        for (int i = 0; i < cnt; i++) {
          const int li = i + kCoderHalo;
          const CoderDescR D = coder_describe(W, T, i, s0 + i, n, bpn, laplace);
          const bool is_sig = !((D.d >> 8) & 1);
          const int idx = (D.a >> 16) & 0xffff;
          CntL cur = csig0[is_sig ? idx : 0], upd = cur;
          const int bit = coder_step(M, T, D, bpn, cur, &upd, rc);
          if (is_sig) csig0[idx] = upd;
          if (bit) {
            W.val[li] += 1 << bpn;
            if (is_sig) W.msb[li] = (unsigned char)bpn;
          }
        }
        W.fl[0] = consumed + rc.pos;
      });
      ex.sync();
      consumed = W.fl[0];
      ex.par([&](int l) { if (l < cnt) s2u[s0 + l] = W.val[l + kCoderHalo]; });
      ex.gsync();
    }
  }
  return consumed;
}

}  // namespace sacamd
