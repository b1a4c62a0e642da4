// sac_amd/csrc/framecoder.h -- FrameCoder-shaped C++ host wrapper over the C ABI (include/sac_amd.h).
//
// Mirrors the public surface of the reference's FrameCoder (/root/reference/src/libsac/libsac.h:12-83)
// for the ENCODE path so that Codec::EncodeFile (libsac/libsac.cpp:782-855) can drive it unchanged:
//   FrameCoder(numchannels, framesize, cfg); samples[ch][0..n); SetNumSamples(n); Predict();
//   Encode(); WriteEncoded(fout) -- same names, same argument meaning, same public buffers.
// One frame per Predict()/Encode() pair, like the reference; base_profile is carried from frame to frame
// unless cfg.ocfg.reset (libsac.cpp:461-466), so a caller that encodes a file frame by frame gets the
// reference's warm-started searches.  Callers that want many frames in one GPU batch use the C ABI's
// sacamd_encode_frames directly.
// Errors: the reference prints and continues or terminates; here every failure throws
// std::runtime_error with sacamd_last_error() (never across the C ABI, which returns codes).
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sac_amd.h"

namespace sacamd {

class FrameCoder {
 public:
  enum SearchCost { L1, RMS, Entropy, Golomb, Bitplane };   // libsac.h:14
  enum SearchMethod { DDS, DE, CMA };                        // libsac.h:15
  struct toptim_cfg {                                         // libsac.h:19-31 (dds_cfg / de_cfg / cma_cfg are derived from these, cmdline.cpp:221-241)
    int reset = 0; double fraction = 0; int maxnfunc = 0; int num_threads = 0; double sigma = 0.2; int optk = 4;
    SearchMethod optimize_search = DDS;
    SearchCost optimize_cost = Entropy;
  };
  struct tsac_cfg {                                           // libsac.h:32-44
    int optimize = 0, sparse_pcm = 1, zero_mean = 1, max_framelen = 20, verbose_level = 0, mt_mode = 2, adapt_block = 1;
    toptim_cfg ocfg;
  };
  struct FrameStats { int maxbpn = 0, maxbpn_map = 0; bool enc_mapped = false; int32_t blocksize = 0, minval = 0, maxval = 0, mean = 0; };

  FrameCoder(int numchannels, int framesize, const tsac_cfg &sac_cfg, int device = 0)
      : numchannels_(numchannels), framesize_(framesize), numsamples_(0), cfg(sac_cfg) {
    if (sacamd_ctx_create(device, numchannels, framesize, 1, &ctx_) != 0)
      throw std::runtime_error("sacamd_ctx_create failed (no gfx950 device?)");
    samples.assign(numchannels, std::vector<int32_t>(framesize));
    error = s2u_error = pred = samples;
    encoded.resize(numchannels);
    framestats.resize(numchannels);
    base_profile.resize(SACAMD_NUM_COEFS);
    sacamd_default_profile(nullptr, nullptr, base_profile.data());
  }
  ~FrameCoder() { sacamd_ctx_destroy(ctx_); }
  FrameCoder(const FrameCoder &) = delete;

  void SetNumSamples(int nsamples) { numsamples_ = nsamples; }
  int GetNumSamples() { return numsamples_; }

  // reference semantics, one frame: Predict() = analyse + (search) + final pass + S2U
  void Predict() {
    stage_current();
    const sacamd_cfg c = ccfg();
    chk(sacamd_analyse(ctx_, &c));
    run_search_and_final(c);
    chk(sacamd_get_residuals(ctx_, 0, flat(error).data(), flat(pred).data(), flat(s2u_error).data(), maxbpn_));
    unflat();
  }
  void Encode() {
    const sacamd_cfg c = ccfg();
    chk(sacamd_encode(ctx_, &c));
    for (int ch = 0; ch < numchannels_; ch++) {
      int len = 0, mapped = 0, mb = 0;
      chk(sacamd_get_encoded(ctx_, 0, ch, nullptr, 0, &len, &mapped, &mb));
      encoded[ch].resize(len);
      chk(sacamd_get_encoded(ctx_, 0, ch, encoded[ch].data(), len, &len, &mapped, &mb));
      framestats[ch].enc_mapped = mapped; framestats[ch].blocksize = len;
      (mapped ? framestats[ch].maxbpn_map : framestats[ch].maxbpn) = mb;
    }
  }
  // frame record exactly as FrameCoder::WriteEncoded (libsac.cpp:565-578)
  void WriteEncoded(std::ostream &fout) {
    auto put32 = [&](uint32_t v) { char b[4]; for (int i = 0; i < 4; i++) b[i] = (char)(v >> (8 * i)); fout.write(b, 4); };
    put32((uint32_t)numsamples_);
    for (float f : base_profile) { uint32_t ix; std::memcpy(&ix, &f, 4); put32(ix); }
    for (int ch = 0; ch < numchannels_; ch++) {
      const FrameStats &st = framestats[ch];
      put32((uint32_t)st.blocksize); put32((uint32_t)st.mean); put32((uint32_t)st.minval); put32((uint32_t)st.maxval);
      const uint16_t flag = st.enc_mapped ? (uint16_t)((1u << 9) | st.maxbpn_map) : (uint16_t)st.maxbpn;
      char b[2] = {(char)flag, (char)(flag >> 8)};
      fout.write(b, 2);
      fout.write(reinterpret_cast<const char *>(encoded[ch].data()), st.blocksize);
    }
  }

  std::vector<std::vector<int32_t>> samples, error, s2u_error, pred;   // public buffers, libsac.h:54
  std::vector<std::vector<uint8_t>> encoded;                           // BufIO payloads
  std::vector<FrameStats> framestats;
  std::vector<float> base_profile;                                     // 58 coefficients (vdef)

 private:
  sacamd_cfg ccfg() const {
    sacamd_cfg c; sacamd_default_cfg(&c);
    c.optimize = cfg.optimize; c.sparse_pcm = cfg.sparse_pcm; c.zero_mean = cfg.zero_mean; c.reset = cfg.ocfg.reset;
    c.fraction = cfg.ocfg.fraction; c.maxnfunc = cfg.ocfg.maxnfunc; c.num_threads = cfg.ocfg.num_threads; c.sigma = cfg.ocfg.sigma;
    c.optk = cfg.ocfg.optk; c.optimize_cost = (int)cfg.ocfg.optimize_cost; c.optimize_search = (int)cfg.ocfg.optimize_search;
    return c;
  }
  void chk(int rc) { if (rc != 0) throw std::runtime_error(std::string("sac_amd: ") + sacamd_last_error(ctx_)); }
  void stage_current() {
    std::vector<int32_t> buf((size_t)numchannels_ * numsamples_);
    for (int ch = 0; ch < numchannels_; ch++) std::memcpy(&buf[(size_t)ch * numsamples_], samples[ch].data(), sizeof(int32_t) * numsamples_);
    chk(sacamd_frames_upload_i32(ctx_, 1, framesize_, buf.data(), (long long)numchannels_ * numsamples_, numsamples_, &numsamples_));
  }
  void run_search_and_final(const sacamd_cfg &c) {
    // FrameCoder::Predict (libsac.cpp:443-479): Optimize (the DDS search) -> base_profile, then the final pass.
    // base_profile is the previous frame's optimum unless --opt-reset (libsac.cpp:461-466: LoadBaseProfile only
    // `if (cfg.ocfg.reset)`; sacamd_search_frames ignores the input in that case); without optimize it is used as is.
    if (c.optimize) chk(sacamd_search_frames(ctx_, &c, base_profile.data()));
    chk(sacamd_predict_final(ctx_, &c, base_profile.data()));
    int32_t st[8];
    chk(sacamd_get_stats(ctx_, st));
    for (int ch = 0; ch < numchannels_; ch++) { framestats[ch].mean = st[4 * ch]; framestats[ch].minval = st[4 * ch + 1]; framestats[ch].maxval = st[4 * ch + 2]; }
  }
  std::vector<int32_t> &flat(std::vector<std::vector<int32_t>> &v) {
    std::vector<int32_t> &f = (&v == &error) ? ferr_ : (&v == &pred ? fpred_ : fs2u_);
    f.resize((size_t)numchannels_ * numsamples_);
    return f;
  }
  void unflat() {
    for (int ch = 0; ch < numchannels_; ch++) {
      std::memcpy(error[ch].data(), &ferr_[(size_t)ch * numsamples_], sizeof(int32_t) * numsamples_);
      std::memcpy(pred[ch].data(), &fpred_[(size_t)ch * numsamples_], sizeof(int32_t) * numsamples_);
      std::memcpy(s2u_error[ch].data(), &fs2u_[(size_t)ch * numsamples_], sizeof(int32_t) * numsamples_);
      framestats[ch].maxbpn = maxbpn_[ch];
    }
  }
  int numchannels_, framesize_, numsamples_;
  tsac_cfg cfg;
  sacamd_ctx *ctx_ = nullptr;
  std::vector<int32_t> ferr_, fpred_, fs2u_;
  int maxbpn_[2] = {0, 0};
};

}  // namespace sacamd
