// sac_amd/csrc/framecoder.h -- FrameCoder-shaped C++ host wrapper over the C ABI (include/sac_amd.h).
//
// Mirrors the public surface of the reference's FrameCoder (/root/reference/src/libsac/libsac.h:12-83) so that
// Codec::EncodeFile (libsac/libsac.cpp:782-855) and Codec::DecodeFile (:857-883) can drive it unchanged:
//   encode: FrameCoder(numchannels, framesize, cfg); samples[ch][0..n); SetNumSamples(n); Predict(); Encode(); WriteEncoded(fout)
//   decode: ReadEncoded(fin); Decode(); Unpredict(); samples[ch][0..GetNumSamples())
// -- same names, same argument meaning, same public buffers (samples, error, s2u_error, s2u_error_map, pred as
// vector<vector<int32_t>>; encoded / enc_temp1 / enc_temp2 as vector<BufIO>; framestats), and the types the reference's callers
// touch: BufIO (common/bufio.h:7-26), SacProfile with coefs{vmin,vmax,vdef} (libsac/profile.h:61-109; base_profile is public
// here, the reference keeps it private).  WriteEncoded / ReadEncoded take anything with a `file` stream member and
// ReadData / WriteData -- i.e. the reference's AudioFile (file/file.h:10-36) as it is -- or a plain std::ostream / std::istream.
// One frame per Predict()/Encode() pair, like the reference; base_profile is carried from frame to frame unless cfg.ocfg.reset
// (libsac.cpp:461-466), so a caller that encodes a file frame by frame gets the reference's warm-started searches.  Callers
// that want many frames in one GPU batch use the C ABI's sacamd_encode_frames / sacamd_decode_frames directly.
// Decode side: the entropy decoder and the un-predictor run as ONE GPU call (sacamd_decode_frames: the decoder's three
// predictor stages chase each other sample by sample, include/sac_amd.h); Decode() runs it, Unpredict() publishes `samples`.
// Errors: the reference prints and continues or terminates; here every failure throws std::runtime_error with
// sacamd_last_error() (never across the C ABI, which returns codes).
// Predictor (libsac/pred.h:9-42) is mirrored in predictor.h (round 5: the encoder-side protocol, replayed from whole-frame GPU
// streams); FrameCoder::SetParam (libsac.cpp:37-92) below builds its tparam from a profile as the reference does.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/sac_amd.h"
#include "predictor.h"

namespace sacamd {

// common/bufio.h:7-26
class BufIO {
 public:
  BufIO() : buf(1024) { Reset(); }
  explicit BufIO(int initsize) : buf(initsize) { Reset(); }
  void Reset() { bufpos = 0; }
  void PutByte(int val) { if (bufpos >= buf.size()) buf.resize(buf.size() * 2); buf[bufpos++] = (uint8_t)val; }
  int GetByte() { return bufpos >= buf.size() ? -1 : buf[bufpos++]; }
  size_t GetBufPos() { return bufpos; }
  std::vector<uint8_t> &GetBuf() { return buf; }
  void assign(const uint8_t *p, size_t n) { if (buf.size() < n) buf.resize(n); std::memcpy(buf.data(), p, n); bufpos = n; }   // (not in the reference)
 private:
  size_t bufpos;
  std::vector<uint8_t> buf;
};

// libsac/profile.h:61-109 (FrameStats without the Remap member: the used-value flags live on the device)
class SacProfile {
 public:
  struct FrameStats { int maxbpn = 0, maxbpn_map = 0; bool enc_mapped = false; int32_t blocksize = 0, minval = 0, maxval = 0, mean = 0; };
  struct coef { float vmin, vmax, vdef; };
  void Init(int numcoefs) { coefs.resize(numcoefs); }
  int LoadBaseProfile() {                                   // profile.cpp:3-89
    float lo[SACAMD_NUM_COEFS], hi[SACAMD_NUM_COEFS], def[SACAMD_NUM_COEFS];
    sacamd_default_profile(lo, hi, def);
    coefs.resize(SACAMD_NUM_COEFS);
    for (int i = 0; i < SACAMD_NUM_COEFS; i++) coefs[i] = coef{lo[i], hi[i], def[i]};
    return 0;
  }
  std::size_t get_size() { return coefs.size(); }
  void Set(int num, double vmin, double vmax, double vdef) { if (num >= 0 && num < (int)coefs.size()) coefs[num] = coef{(float)vmin, (float)vmax, (float)vdef}; }
  float Get(std::size_t num) const { return num < coefs.size() ? coefs[num].vdef : 0.f; }
  std::vector<coef> coefs;
};

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
  using FrameStats = SacProfile::FrameStats;

  // tsac_cfg -> the C ABI's flat sacamd_cfg (what INTEGRATION.md's batch driver passes to sacamd_encode_frames)
  static sacamd_cfg to_sacamd_cfg(const tsac_cfg &cfg) {
    sacamd_cfg c; sacamd_default_cfg(&c);
    c.optimize = cfg.optimize; c.sparse_pcm = cfg.sparse_pcm; c.zero_mean = cfg.zero_mean; c.reset = cfg.ocfg.reset;
    c.fraction = cfg.ocfg.fraction; c.maxnfunc = cfg.ocfg.maxnfunc; c.num_threads = cfg.ocfg.num_threads; c.sigma = cfg.ocfg.sigma;
    c.optk = cfg.ocfg.optk; c.optimize_cost = (int)cfg.ocfg.optimize_cost; c.optimize_search = (int)cfg.ocfg.optimize_search;
    return c;
  }

  // FrameCoder::SetParam (libsac.cpp:37-92): profile -> Predictor::tparam (k = optk while optimising, 1 for the final pass / decoder)
  static void SetParam(Predictor::tparam &param, const SacProfile &profile, bool optimize, int optk = 4) {
    auto G = [&](std::size_t i) { return (double)profile.Get(i); };
    auto R = [&](std::size_t i) { return (int)std::round((double)profile.Get(i)); };
    param.k = optimize ? optk : 1;
    param.lambda0 = G(0); param.ols_nu0 = G(1);
    param.vn0 = {R(28), R(29), R(30), R(37)};
    param.vn1 = {R(31), R(32), R(33), R(38)};
    param.vmu0 = {G(2) / double(param.vn0[0]), G(3) / double(param.vn0[1]), G(4) / double(param.vn0[2]), G(5) / double(param.vn0[3])};
    param.vmudecay0 = {G(6), G(39), G(46), G(47)};
    param.vpowdecay0 = {G(7), G(8), G(50), G(51)};
    param.mu_mix0 = G(10); param.mu_mix_beta0 = G(11);
    param.lambda1 = G(12); param.ols_nu1 = G(13);
    param.vmu1 = {G(14) / double(param.vn1[0]), G(15) / double(param.vn1[1]), G(16) / double(param.vn1[2]), G(17) / double(param.vn1[3])};
    param.vmudecay1 = {G(18), G(40), G(48), G(49)};
    param.vpowdecay1 = {G(19), G(20), G(21), G(52)};
    param.mu_mix1 = G(22); param.mu_mix_beta1 = G(23);
    param.nA = R(24); param.nB = R(25); param.nS0 = R(26); param.nS1 = R(27); param.nM0 = R(9);
    param.beta_sum0 = G(34); param.beta_pow0 = G(35); param.beta_add0 = G(36);
    param.beta_sum1 = G(53); param.beta_pow1 = G(54); param.beta_add1 = G(55);
    param.proj_alpha0 = G(56); param.proj_alpha1 = G(57);
    param.lm_n = R(41); param.lm_alpha = G(42);
    param.bias_mu0 = G(43); param.bias_mu1 = G(44);
    param.bias_scale0 = param.bias_scale1 = R(45);
    param.ch_ref = 0;
    if (param.nS1 < 0) { param.nS1 = -param.nS1; param.ch_ref = 1; }
  }

  FrameCoder(int numchannels, int framesize, const tsac_cfg &sac_cfg, int device = 0)
      : numchannels_(numchannels), framesize_(framesize), numsamples_(0), cfg(sac_cfg) {
    if (sacamd_ctx_create(device, numchannels, framesize, 1, &ctx_) != 0)
      throw std::runtime_error("sacamd_ctx_create failed (no gfx950 device?)");
    samples.assign(numchannels, std::vector<int32_t>(framesize));
    error = s2u_error = s2u_error_map = pred = samples;
    encoded.resize(numchannels); enc_temp1.resize(numchannels); enc_temp2.resize(numchannels);
    framestats.resize(numchannels);
    base_profile.LoadBaseProfile();
    profile_size_bytes_ = (int)base_profile.get_size() * 4;
  }
  ~FrameCoder() { sacamd_ctx_destroy(ctx_); }
  FrameCoder(const FrameCoder &) = delete;

  void SetNumSamples(int nsamples) { numsamples_ = nsamples; }
  int GetNumSamples() { return numsamples_; }

  // reference semantics, one frame: Predict() = analyse + (search) + final pass + S2U
  void Predict() {
    stage_current();
    const sacamd_cfg c = to_sacamd_cfg(cfg);
    chk(sacamd_analyse(ctx_, &c));
    run_search_and_final(c);
    chk(sacamd_get_residuals(ctx_, 0, flat(error).data(), flat(pred).data(), flat(s2u_error).data(), maxbpn_));
    unflat();
  }
  void Encode() {
    const sacamd_cfg c = to_sacamd_cfg(cfg);
    chk(sacamd_encode(ctx_, &c));
    std::vector<uint8_t> tmp;
    for (int ch = 0; ch < numchannels_; ch++) {
      int len = 0, mapped = 0, mb = 0;
      chk(sacamd_get_encoded(ctx_, 0, ch, nullptr, 0, &len, &mapped, &mb));
      tmp.resize((size_t)len + 1);
      chk(sacamd_get_encoded(ctx_, 0, ch, tmp.data(), len, &len, &mapped, &mb));
      encoded[ch].assign(tmp.data(), (size_t)len);
      framestats[ch].enc_mapped = mapped != 0; framestats[ch].blocksize = len;
      (mapped ? framestats[ch].maxbpn_map : framestats[ch].maxbpn) = mb;
      // what EncodeMonoFrame leaves beside `encoded` (libsac.cpp:253-278): enc_temp1 = the Normal stream, enc_temp2 = the Mapped one (if it was tried)
      for (int variant = 0; variant < 2; variant++) {
        BufIO &dst = variant ? enc_temp2[ch] : enc_temp1[ch];
        int vl = 0, vmb = 0;
        chk(sacamd_get_encoded_variant(ctx_, 0, ch, variant, nullptr, 0, &vl, &vmb));
        dst.Reset();
        if (vl > 0) { tmp.resize((size_t)vl + 1); chk(sacamd_get_encoded_variant(ctx_, 0, ch, variant, tmp.data(), vl, &vl, &vmb)); dst.assign(tmp.data(), (size_t)vl); }
      }
    }
    if (c.sparse_pcm) {     // CalcRemapError's products (libsac.cpp:230-251): s2u_error_map, framestats[].maxbpn_map
      std::vector<int32_t> m((size_t)numchannels_ * numsamples_);
      int mbm[2] = {0, 0};
      chk(sacamd_get_residuals_map(ctx_, 0, m.data(), mbm));
      for (int ch = 0; ch < numchannels_; ch++) {
        std::memcpy(s2u_error_map[ch].data(), &m[(size_t)ch * numsamples_], sizeof(int32_t) * (size_t)numsamples_);
        framestats[ch].maxbpn_map = mbm[ch];
      }
    }
  }

  // ---- frame record I/O, exactly FrameCoder::WriteEncoded / ReadEncoded (libsac.cpp:565-594)
  // AudioFile-shaped target: anything with a `file` stream member (file/file.h:33)
  template <class AF, class = decltype(std::declval<AF &>().file)>
  void WriteEncoded(AF &fout) { write_record(fout.file); }
  void WriteEncoded(std::ostream &fout) { write_record(fout); }
  template <class AF, class = decltype(std::declval<AF &>().file)>
  void ReadEncoded(AF &fin) { read_record(fin.file); }
  void ReadEncoded(std::istream &fin) { read_record(fin); }

  // FrameCoder::WriteBlockHeader / ReadBlockHeader (libsac.cpp:530-563), on any stream
  template <class S> static int WriteBlockHeader(S &file, const std::vector<FrameStats> &framestats, int ch) {
    uint8_t buf[18];
    put32(buf, (uint32_t)framestats[ch].blocksize); put32(buf + 4, (uint32_t)framestats[ch].mean);
    put32(buf + 8, (uint32_t)framestats[ch].minval); put32(buf + 12, (uint32_t)framestats[ch].maxval);
    const uint16_t flag = framestats[ch].enc_mapped ? (uint16_t)((1u << 9) | framestats[ch].maxbpn_map) : (uint16_t)framestats[ch].maxbpn;
    buf[16] = (uint8_t)flag; buf[17] = (uint8_t)(flag >> 8);
    file.write(reinterpret_cast<const char *>(buf), 18);
    return 18;
  }
  template <class S> static int ReadBlockHeader(S &file, std::vector<FrameStats> &framestats, int ch) {
    uint8_t buf[18];
    file.read(reinterpret_cast<char *>(buf), 18);
    framestats[ch].blocksize = (int32_t)get32(buf); framestats[ch].mean = (int32_t)get32(buf + 4);
    framestats[ch].minval = (int32_t)get32(buf + 8); framestats[ch].maxval = (int32_t)get32(buf + 12);
    const uint16_t flag = (uint16_t)(buf[16] | (buf[17] << 8));
    framestats[ch].enc_mapped = (flag >> 9) != 0;
    framestats[ch].maxbpn = flag & 0xff;                     // (as the reference: the plane count lands in maxbpn for both kinds)
    return 18;
  }

  // ---- decode side (libsac.cpp:496-505, 144-199): Decode() = entropy decode + un-predict of the record ReadEncoded took in,
  // as one GPU call; Unpredict() hands the PCM to `samples` (mean added back, libsac.cpp:194-198)
  void Decode() {
    if (numsamples_ < 1 || numsamples_ > framesize_) throw std::runtime_error("FrameCoder::Decode: no frame read");
    std::vector<uint8_t> rec;
    auto push32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) rec.push_back((uint8_t)(v >> (8 * i))); };
    push32((uint32_t)numsamples_);
    for (auto &c : base_profile.coefs) { uint32_t ix; std::memcpy(&ix, &c.vdef, 4); push32(ix); }
    for (int ch = 0; ch < numchannels_; ch++) {
      const FrameStats &st = framestats[ch];
      push32((uint32_t)st.blocksize); push32((uint32_t)st.mean); push32((uint32_t)st.minval); push32((uint32_t)st.maxval);
      const uint16_t flag = st.enc_mapped ? (uint16_t)((1u << 9) | st.maxbpn) : (uint16_t)st.maxbpn;    // ReadBlockHeader keeps the planes in maxbpn
      rec.push_back((uint8_t)flag); rec.push_back((uint8_t)(flag >> 8));
      rec.insert(rec.end(), encoded[ch].GetBuf().begin(), encoded[ch].GetBuf().begin() + st.blocksize);
    }
    const long long off[2] = {0, (long long)rec.size()};
    decoded_.assign((size_t)numchannels_ * framesize_, 0);
    int n = 0;
    chk(sacamd_decode_frames(ctx_, 1, framesize_, rec.data(), off, decoded_.data(), (long long)numchannels_ * framesize_, framesize_, &n, nullptr));
    if (n != numsamples_) throw std::runtime_error("FrameCoder::Decode: sample count of the record changed");
  }
  void Unpredict() {
    if (decoded_.size() != (size_t)numchannels_ * framesize_) throw std::runtime_error("FrameCoder::Unpredict before Decode");
    for (int ch = 0; ch < numchannels_; ch++) std::memcpy(samples[ch].data(), &decoded_[(size_t)ch * framesize_], sizeof(int32_t) * (size_t)numsamples_);
  }

  std::vector<std::vector<int32_t>> samples, error, s2u_error, s2u_error_map, pred;   // public buffers, libsac.h:54
  std::vector<BufIO> encoded, enc_temp1, enc_temp2;                                    // libsac.h:55 (enc_temp1 / enc_temp2: the Normal / Mapped stream after Encode(), as in the reference)
  std::vector<FrameStats> framestats;
  SacProfile base_profile;                                                             // 58 coefficients; .coefs[i].vdef is what a frame record stores

 private:
  static void put32(uint8_t *b, uint32_t v) { for (int i = 0; i < 4; i++) b[i] = (uint8_t)(v >> (8 * i)); }
  static uint32_t get32(const uint8_t *b) { return (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24); }
  template <class S> void write_record(S &file) {
    uint8_t buf[4];
    put32(buf, (uint32_t)numsamples_);
    file.write(reinterpret_cast<const char *>(buf), 4);
    std::vector<uint8_t> profile_buf(profile_size_bytes_);
    for (size_t i = 0; i < base_profile.coefs.size(); i++) { uint32_t ix; std::memcpy(&ix, &base_profile.coefs[i].vdef, 4); put32(&profile_buf[4 * i], ix); }   // EncodeProfile, libsac.cpp:507-518
    file.write(reinterpret_cast<const char *>(profile_buf.data()), profile_size_bytes_);
    for (int ch = 0; ch < numchannels_; ch++) {
      framestats[ch].blocksize = (int32_t)encoded[ch].GetBufPos();
      WriteBlockHeader(file, framestats, ch);
      file.write(reinterpret_cast<const char *>(encoded[ch].GetBuf().data()), framestats[ch].blocksize);
    }
  }
  template <class S> void read_record(S &file) {
    uint8_t buf[4];
    file.read(reinterpret_cast<char *>(buf), 4);
    numsamples_ = (int)get32(buf);
    std::vector<uint8_t> profile_buf(profile_size_bytes_);
    file.read(reinterpret_cast<char *>(profile_buf.data()), profile_size_bytes_);
    for (size_t i = 0; i < base_profile.coefs.size(); i++) { const uint32_t ix = get32(&profile_buf[4 * i]); std::memcpy(&base_profile.coefs[i].vdef, &ix, 4); }   // DecodeProfile, libsac.cpp:520-528
    for (int ch = 0; ch < numchannels_; ch++) {
      ReadBlockHeader(file, framestats, ch);
      std::vector<uint8_t> &b = encoded[ch].GetBuf();
      if (b.size() < (size_t)framestats[ch].blocksize) b.resize((size_t)framestats[ch].blocksize);
      file.read(reinterpret_cast<char *>(b.data()), framestats[ch].blocksize);        // AudioFile::ReadData, file/file.cpp
    }
    if (!file) throw std::runtime_error("FrameCoder::ReadEncoded: short read");
    decoded_.clear();
  }
  void chk(int rc) { if (rc != 0) throw std::runtime_error(std::string("sac_amd: ") + sacamd_last_error(ctx_)); }
  void stage_current() {
    std::vector<int32_t> buf((size_t)numchannels_ * numsamples_);
    for (int ch = 0; ch < numchannels_; ch++) std::memcpy(&buf[(size_t)ch * numsamples_], samples[ch].data(), sizeof(int32_t) * numsamples_);
    chk(sacamd_frames_upload_i32(ctx_, 1, framesize_, buf.data(), (long long)numchannels_ * numsamples_, numsamples_, &numsamples_));
  }
  void run_search_and_final(const sacamd_cfg &c) {
    // FrameCoder::Predict (libsac.cpp:443-479): Optimize (the search) -> base_profile, then the final pass.
    // base_profile is the previous frame's optimum unless --opt-reset (libsac.cpp:461-466: LoadBaseProfile only
    // `if (cfg.ocfg.reset)`; sacamd_search_frames ignores the input in that case); without optimize it is used as is.
    std::vector<float> prof(base_profile.coefs.size());
    for (size_t i = 0; i < prof.size(); i++) prof[i] = base_profile.coefs[i].vdef;
    if (c.optimize) chk(sacamd_search_frames(ctx_, &c, prof.data()));
    chk(sacamd_predict_final(ctx_, &c, prof.data()));
    for (size_t i = 0; i < prof.size(); i++) base_profile.coefs[i].vdef = prof[i];
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
  int profile_size_bytes_ = 0;
  tsac_cfg cfg;
  sacamd_ctx *ctx_ = nullptr;
  std::vector<int32_t> ferr_, fpred_, fs2u_, decoded_;
  int maxbpn_[2] = {0, 0};
};

}  // namespace sacamd
