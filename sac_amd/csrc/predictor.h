// sac_amd/csrc/predictor.h -- the reference's Predictor class surface (/root/reference/src/libsac/pred.h:9-42, pred.cpp:4-46) for
// the ENCODER, over the C ABI (include/sac_amd.h: sacamd_predictor_streams).
//
//   Predictor(Range r0, Range r1, const tparam &p);  fillbuf_ch0 / fillbuf_ch1;  double predict(int ch);  void update(int ch, double val);
//   public members p, nA, nB, nM0, nS0, nS1, p_lpc[2], p_lms[2]
// -- same names, same argument meaning, so FrameCoder::PredictFrame's loop (libsac.cpp:113-141) compiles against it unchanged.
//
// How it differs inside.  The reference's object advances one sample per predict()/update() pair on the CPU.  In the encoder every
// sample of the frame is known in advance (fillbuf_* receive the frame's base pointers), so the three recurrences behind predict()
// -- OLS, cascade, bias -- run over the WHOLE frame on the GPU at the first predict() and the calls that follow replay
// the streams: predict(ch) returns the prediction of the sample the last fillbuf_ch<ch> pointed at, update(ch, val) checks that val
// is that sample (anything else is not the encoder's protocol: std::logic_error) and steps on.  pd and p_lpc are the reference's to
// the last bit (tests: against the genuine reference's traces); the public p_lms[] is (p_lpc + p_lms) - p_lpc of the cascade kernel's
// output stream, i.e. the reference's value up to one rounding of that difference -- nothing in PredictFrame reads it.
// The frame length.  The whole-frame replay needs `numsamples`, which the reference's constructor does not carry:
//   * Predictor(r0, r1, param)            -- the reference's signature (libsac.cpp:102 compiles unchanged).  Stereo: the length arrives
//     with the first fillbuf_ch1 (pred.cpp:25); the at most nS1 predict(0) calls before it (libsac.cpp:127-140) are answered from
//     causal prefixes of the frame (a prediction at t only depends on samples before t).  Mono never calls fillbuf_ch1: give the
//     length with Predictor::frame_length_hint(numsamples) before constructing (thread-local, consumed by the next constructor),
//     otherwise every predict() recomputes the prefix up to its sample and the 65th such call throws.
//   * Predictor(r0, r1, param, numsamples[, device]) -- the length up front.
// fillbuf_ch0's idx1 must follow PredictFrame's stereo schedule idx1 = max(0, idx0 - (max(nS1, 1) - 1)) (libsac.cpp:127-140), which
// is what the whole-frame OLS stage of channel 0 assumes (params.h: du); any other driver gets std::logic_error, not other numbers.
// The decoder cannot use this class -- there a sample only exists after it has been predicted; sacamd::FrameCoder::Decode /
// sacamd_decode_frames are the decode side (framecoder.h).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sac_amd.h"

namespace sacamd {

struct Range { int32_t lo, hi; };      // pred/cascade.h:71-73

class Predictor {
 public:
  struct tparam {                      // libsac/pred.h:11-27
    int nA, nB, nM0, nS0, nS1, k;
    std::vector<int> vn0, vn1;
    std::vector<double> vmu0, vmu1;
    std::vector<double> vmudecay0, vmudecay1;
    std::vector<double> vpowdecay0, vpowdecay1;
    double lambda0, lambda1, ols_nu0, ols_nu1, mu_mix0, mu_mix1, mu_mix_beta0, mu_mix_beta1;
    double beta_sum0, beta_pow0, beta_add0;
    double beta_sum1, beta_pow1, beta_add1;
    int ch_ref;
    double bias_mu0, bias_mu1;
    int bias_scale0, bias_scale1;
    int lm_n;
    double lm_alpha;
    double proj_alpha0, proj_alpha1;
  };
  // the reference's constructor (pred.h:28): the frame length comes from frame_length_hint() or from the first fillbuf_ch1
  Predictor(Range r0, Range r1, const tparam &p)
      : p(p), nA(p.nA), nB(p.nB), nM0(p.nM0), nS0(p.nS0), nS1(p.nS1), r0_(r0), r1_(r1), n_(hint_slot()), device_(0) {
    for (int i = 0; i < 2; i++) p_lpc[i] = p_lms[i] = 0.0;
    hint_slot() = -1;
  }
  // numsamples: length of the frame (window) the caller is about to walk; device: HIP device ordinal
  Predictor(Range r0, Range r1, const tparam &p, int numsamples, int device = 0)
      : p(p), nA(p.nA), nB(p.nB), nM0(p.nM0), nS0(p.nS0), nS1(p.nS1), r0_(r0), r1_(r1), n_(numsamples), device_(device) {
    for (int i = 0; i < 2; i++) p_lpc[i] = p_lms[i] = 0.0;
    if (numsamples < 1) throw std::invalid_argument("sacamd::Predictor: numsamples < 1");
  }
  // frame length for the NEXT three-argument construction on this thread (mono callers; see the header comment)
  static void frame_length_hint(int numsamples) { hint_slot() = numsamples > 0 ? numsamples : -1; }
  ~Predictor() { if (ctx_) sacamd_ctx_destroy(ctx_); }
  Predictor(const Predictor &) = delete;

  void fillbuf_ch0(const int32_t *src0, int idx0, const int32_t *src1, int idx1) {       // pred.cpp:17-23
    bind(src0, src1);
    if (src1 != src0) {                    // stereo: the schedule the whole-frame OLS stage of channel 0 assumes (libsac.cpp:127-140)
      const int lag = (nS1 > 1 ? nS1 : 1) - 1, want = idx0 > lag ? idx0 - lag : 0;
      if (idx1 != want) throw std::logic_error("sacamd::Predictor::fillbuf_ch0: idx1 is not PredictFrame's schedule max(0, idx0 - (max(nS1, 1) - 1))");
    }
    idx_[0] = idx0;
  }
  void fillbuf_ch1(const int32_t *src0, const int32_t *src1, int idx1, int numsamples) {  // pred.cpp:25-31
    bind(src0, src1);
    if (n_ < 0) {                          // three-argument constructor: this is where the frame length arrives
      if (numsamples < 1) throw std::invalid_argument("sacamd::Predictor: numsamples < 1");
      n_ = numsamples; have_ = 0;
    } else if (numsamples != n_) throw std::logic_error("sacamd::Predictor: fillbuf_ch1 with another frame length than the constructor's");
    idx_[1] = idx1;
  }
  double predict(int ch) {                                                                // pred.cpp:33-38
    if (n_ > 0) { if (have_ < n_) run(n_); }
    else {                                 // length not known yet: the causal prefix that ends with this sample
      if (ch != 0 || !src0_) throw std::logic_error("sacamd::Predictor: predict(1) before fillbuf_ch1");
      if (idx_[0] < 0) throw std::out_of_range("sacamd::Predictor: sample index outside the frame");
      if (have_ <= idx_[0]) {
        if (++prefix_runs_ > 64) throw std::logic_error("sacamd::Predictor: frame length unknown (mono): call Predictor::frame_length_hint(numsamples) before "
                                                        "constructing, or use the four-argument constructor");
        run(idx_[0] + 1);
      }
    }
    const int t = at(ch);
    p_lpc[ch] = plpc_[(size_t)ch * len_ + t];
    p_lms[ch] = plms_[(size_t)ch * len_ + t];
    return pd_[(size_t)ch * len_ + t];
  }
  void update(int ch, double val) {                                                       // pred.cpp:40-46
    const int t = at(ch);
    const int32_t *s = ch == 0 ? src0_ : src1_;
    if ((double)s[t] != val) throw std::logic_error("sacamd::Predictor::update: not the frame's own sample (encoder protocol only)");
  }

  tparam p;
  int nA, nB, nM0, nS0, nS1;
  double p_lpc[2], p_lms[2];

 private:
  void bind(const int32_t *s0, const int32_t *s1) {
    if (!src0_) { src0_ = s0; src1_ = s1; }
    else if (src0_ != s0 || src1_ != s1) throw std::logic_error("sacamd::Predictor: the frame's base pointers changed");
  }
  int at(int ch) const {
    if (ch < 0 || ch > 1 || !src0_) throw std::logic_error("sacamd::Predictor: predict / update before fillbuf");
    const int t = idx_[ch];
    if (t < 0 || t >= have_) throw std::out_of_range("sacamd::Predictor: sample index outside the frame");
    return t;
  }
  static int &hint_slot() { static thread_local int h = -1; return h; }
  // the three stage recurrences over samples [0, len) of the frame
  void run(int len) {
    const int nch = (src1_ == src0_) ? 1 : 2;              // mono passes its own signal as "other channel" (libsac.cpp:117)
    const int cap = n_ > 0 ? n_ : 4096;                    // context capacity: the frame when known, else room for the short prefixes
    if (ctx_ && (ctx_cap_ < len || ctx_nch_ != nch)) { sacamd_ctx_destroy(ctx_); ctx_ = nullptr; }
    if (!ctx_) {
      ctx_cap_ = len > cap ? len : cap; ctx_nch_ = nch;
      if (sacamd_ctx_create(device_, nch, ctx_cap_ < 16 ? 16 : ctx_cap_, 1, &ctx_) != 0) throw std::runtime_error("sacamd_ctx_create failed (no gfx950 device?)");
    }
    sacamd_pred_tparam t;
    t.nA = p.nA; t.nB = p.nB; t.nM0 = p.nM0; t.nS0 = p.nS0; t.nS1 = p.nS1; t.k = p.k;
    auto need4 = [](size_t n) { if (n != 4) throw std::invalid_argument("sacamd::Predictor: four cascade stages (vn / vmu / vmudecay / vpowdecay)"); };
    need4(p.vn0.size()); need4(p.vmu0.size()); need4(p.vmudecay0.size()); need4(p.vpowdecay0.size());
    if (nch == 2) { need4(p.vn1.size()); need4(p.vmu1.size()); need4(p.vmudecay1.size()); need4(p.vpowdecay1.size()); }
    for (int i = 0; i < 4; i++) {
      t.vn0[i] = p.vn0[i]; t.vmu0[i] = p.vmu0[i]; t.vmudecay0[i] = p.vmudecay0[i]; t.vpowdecay0[i] = p.vpowdecay0[i];
      const bool s1 = nch == 2;
      t.vn1[i] = s1 ? p.vn1[i] : p.vn0[i]; t.vmu1[i] = s1 ? p.vmu1[i] : p.vmu0[i];
      t.vmudecay1[i] = s1 ? p.vmudecay1[i] : p.vmudecay0[i]; t.vpowdecay1[i] = s1 ? p.vpowdecay1[i] : p.vpowdecay0[i];
    }
    t.lambda0 = p.lambda0; t.lambda1 = p.lambda1; t.ols_nu0 = p.ols_nu0; t.ols_nu1 = p.ols_nu1;
    t.mu_mix0 = p.mu_mix0; t.mu_mix1 = p.mu_mix1; t.mu_mix_beta0 = p.mu_mix_beta0; t.mu_mix_beta1 = p.mu_mix_beta1;
    t.beta_sum0 = p.beta_sum0; t.beta_pow0 = p.beta_pow0; t.beta_add0 = p.beta_add0;
    t.beta_sum1 = p.beta_sum1; t.beta_pow1 = p.beta_pow1; t.beta_add1 = p.beta_add1;
    t.ch_ref = p.ch_ref; t.bias_mu0 = p.bias_mu0; t.bias_mu1 = p.bias_mu1; t.bias_scale0 = p.bias_scale0; t.bias_scale1 = p.bias_scale1;
    t.lm_n = p.lm_n; t.lm_alpha = p.lm_alpha; t.proj_alpha0 = p.proj_alpha0; t.proj_alpha1 = p.proj_alpha1;
    const int32_t r4[4] = {r0_.lo, r0_.hi, r1_.lo, r1_.hi};
    pd_.assign((size_t)nch * len, 0.0); plpc_ = pd_; plms_ = pd_;
    if (sacamd_predictor_streams(ctx_, nch, src0_, src1_, len, r4, &t, pd_.data(), plpc_.data(), plms_.data()) != 0)
      throw std::runtime_error(std::string("sac_amd: ") + sacamd_last_error(ctx_));
    have_ = len_ = len;
  }
  Range r0_, r1_;
  int n_, device_;                          // n_ < 0: frame length not known yet
  const int32_t *src0_ = nullptr, *src1_ = nullptr;
  int idx_[2] = {0, 0};
  int have_ = 0, len_ = 0;                  // samples the streams below cover / their per-channel stride
  int prefix_runs_ = 0, ctx_cap_ = 0, ctx_nch_ = 0;
  sacamd_ctx *ctx_ = nullptr;
  std::vector<double> pd_, plpc_, plms_;
};

}  // namespace sacamd
