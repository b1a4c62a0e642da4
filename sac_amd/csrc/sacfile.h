// sac_amd/csrc/sacfile.h -- WAV / .sac container layer in C++ (host only), the counterpart of
// sac_amd/container.py for C++ callers.  Formats and quirks as the reference has them:
//   Wav::ReadHeader / ReadSamples   /root/reference/src/file/wav.cpp:77-125,166-263
//   Chunks::PackMetaData            file/wav.cpp:24-36
//   Sac::WriteSACHeader / WriteMD5  file/sac.cpp:5-38
//   MD5 over the sample bytes       common/md5.cpp (RFC 1321; own implementation below)
// 8- and 16-bit PCM (the scope of the GPU path).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace sacamd {

struct WavChunk { uint32_t id, size; std::vector<uint8_t> payload; };

struct WavInfo {
  int numchannels = 0, samplerate = 0, bitspersample = 0, blockalign = 0, numsamples = 0;
  std::vector<WavChunk> chunks;          // file order, as the reference records them
  std::vector<uint8_t> data;             // sample bytes (numsamples * blockalign)
  size_t metadatasize() const { size_t s = 0; for (auto &c : chunks) s += 8 + c.payload.size(); return s; }
};

inline uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline void wr32(std::vector<uint8_t> &o, uint32_t v) { for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i))); }
inline void wr16(std::vector<uint8_t> &o, uint16_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }
inline size_t word_align(size_t n) { return n + (n & 1); }

constexpr uint32_t kIdRiff = 0x46464952, kIdFmt = 0x20746D66, kIdData = 0x61746164;

inline WavInfo parse_wav(const std::vector<uint8_t> &raw) {
  WavInfo w;
  if (raw.size() < 12 || rd32(&raw[0]) != kIdRiff || rd32(&raw[8]) != 0x45564157) throw std::runtime_error("not a RIFF/WAVE file");
  w.chunks.push_back({kIdRiff, rd32(&raw[4]), std::vector<uint8_t>(raw.begin() + 8, raw.begin() + 12)});
  size_t pos = 12;
  const size_t size = raw.size();
  bool have_data = false;
  while (pos + 8 <= size) {
    const uint32_t cid = rd32(&raw[pos]), csz = rd32(&raw[pos + 4]);
    pos += 8;
    if (cid == kIdFmt) {
      if (csz != 16 && csz != 18 && csz != 40) throw std::runtime_error("invalid fmt chunk size");
      if (pos + csz > size) throw std::runtime_error("truncated fmt chunk");
      const uint8_t *b = &raw[pos];
      w.chunks.push_back({cid, csz, std::vector<uint8_t>(b, b + csz)});
      int fmt = rd16(b);
      w.numchannels = rd16(b + 2); w.samplerate = (int)rd32(b + 4); w.blockalign = rd16(b + 12); w.bitspersample = rd16(b + 14);
      if (csz >= 18 && rd16(b + 16) >= 22) { w.bitspersample = rd16(b + 18); fmt = rd16(b + 24); }
      if (fmt != 1) throw std::runtime_error("only PCM is supported");
      pos += csz;
    } else if (cid == kIdData) {
      if (w.blockalign == 0) throw std::runtime_error("data chunk before fmt chunk");
      w.chunks.push_back({cid, csz, {}});
      have_data = true;
      const size_t end = pos + word_align(csz);
      size_t nbytes = csz;
      if (end >= size) {                            // last chunk: stop here (wav.cpp:233-242)
        if (end > size) nbytes = (size - pos) / w.blockalign * w.blockalign;
        w.numsamples = (int)(nbytes / w.blockalign);
        w.data.assign(raw.begin() + pos, raw.begin() + pos + (size_t)w.numsamples * w.blockalign);
        break;
      }
      w.numsamples = (int)(csz / w.blockalign);
      w.data.assign(raw.begin() + pos, raw.begin() + pos + (size_t)w.numsamples * w.blockalign);
      pos += csz;                                   // not word-aligned in the reference (wav.cpp:243-245)
    } else {
      const size_t n = word_align(csz);
      if (pos + n > size) throw std::runtime_error("truncated chunk");
      w.chunks.push_back({cid, csz, std::vector<uint8_t>(raw.begin() + pos, raw.begin() + pos + n)});
      pos += n;
    }
    if (pos == size) break;
  }
  if (!have_data) throw std::runtime_error("no data chunk");
  return w;
}

// planar int32 samples [ch][n] (wav.cpp:91-108)
inline std::vector<int32_t> pcm_from_wav(const WavInfo &w) {
  const int cs = w.blockalign / w.numchannels, n = w.numsamples, nch = w.numchannels;
  std::vector<int32_t> out((size_t)nch * n);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < nch; k++) {
      const uint8_t *p = &w.data[((size_t)i * nch + k) * cs];
      if (cs == 1) out[(size_t)k * n + i] = (int32_t)p[0] - 128;
      else if (cs == 2) out[(size_t)k * n + i] = (int16_t)(p[0] | (p[1] << 8));
      else if (cs == 3) out[(size_t)k * n + i] = (int32_t)(((uint32_t)p[2] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[0] << 8)) >> 8;   // wav.cpp:109-121
      else throw std::runtime_error("unsupported sample size (8-, 16- and 24-bit PCM)");
    }
  return out;
}

inline std::vector<uint8_t> pack_metadata(const WavInfo &w) {
  std::vector<uint8_t> m;
  for (auto &c : w.chunks) { wr32(m, c.id); wr32(m, c.size); m.insert(m.end(), c.payload.begin(), c.payload.end()); }
  return m;
}

// ---- MD5 (RFC 1321)
struct Md5 {
  uint32_t a = 0x67452301, b = 0xefcdab89, c = 0x98badcfe, d = 0x10325476;
  uint64_t len = 0;
  uint8_t buf[64]; int fill = 0;
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t *p) {
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    static uint32_t K[64]; static bool init = false;
    if (!init) { for (int i = 0; i < 64; i++) { double v = __builtin_fabs(__builtin_sin((double)(i + 1))); K[i] = (uint32_t)(v * 4294967296.0); } init = true; }
    uint32_t M[16]; for (int i = 0; i < 16; i++) M[i] = rd32(p + 4 * i);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; i++) {
      uint32_t F; int g;
      if (i < 16) { F = (B & C) | (~B & D); g = i; }
      else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { F = C ^ (B | ~D); g = (7 * i) & 15; }
      F = F + A + K[i] + M[g];
      A = D; D = C; C = B; B = B + rol(F, S[i]);
    }
    a += A; b += B; c += C; d += D;
  }
  void update(const uint8_t *p, size_t n) {
    len += n;
    while (n) {
      const size_t t = (size_t)(64 - fill) < n ? (size_t)(64 - fill) : n;
      std::memcpy(buf + fill, p, t); fill += (int)t; p += t; n -= t;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void finish(uint8_t out[16]) {
    const uint64_t bits = len * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (8 * i));
    update(lb, 8);
    const uint32_t v[4] = {a, b, c, d};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
  }
};

// header | MD5 | (records follow)
inline std::vector<uint8_t> sac_header_and_md5(const WavInfo &w, int max_framelen) {
  std::vector<uint8_t> h = {'S', 'A', 'C', '2'};
  wr16(h, (uint16_t)w.numchannels); wr32(h, (uint32_t)w.samplerate); wr16(h, (uint16_t)w.bitspersample); wr32(h, (uint32_t)w.numsamples);
  h.push_back((uint8_t)max_framelen); h.push_back(0);
  const std::vector<uint8_t> meta = pack_metadata(w);
  wr32(h, (uint32_t)meta.size());
  h.insert(h.end(), meta.begin(), meta.end());
  Md5 md; md.update(w.data.data(), w.data.size());
  uint8_t dig[16]; md.finish(dig);
  h.insert(h.end(), dig, dig + 16);
  return h;
}

// ---- reading side used by `sacenc --list / --listfull` (cmdline.cpp:295-323, Codec::ScanFrames libsac.cpp:659-693)
struct SacHeader { int numchannels = 0, samplerate = 0, bitspersample = 0, numsamples = 0, max_framelen = 0, metadatasize = 0; uint8_t md5[16] = {}; size_t frames_at = 0; };
inline bool read_sac_header(const std::vector<uint8_t> &raw, SacHeader &h) {
  if (raw.size() < 22 || std::memcmp(raw.data(), "SAC2", 4) != 0) return false;
  h.numchannels = rd16(&raw[4]); h.samplerate = (int)rd32(&raw[6]); h.bitspersample = rd16(&raw[10]); h.numsamples = (int)rd32(&raw[12]);
  h.max_framelen = raw[16]; h.metadatasize = (int)rd32(&raw[18]);
  const size_t p = 22 + (size_t)h.metadatasize;
  if (raw.size() < p + 16) return false;
  std::memcpy(h.md5, &raw[p], 16);
  h.frames_at = p + 16;
  return true;
}
// Chunks::UnpackMetaData (file/wav.cpp) inverse of pack_metadata: RIFF keeps its 4-byte form type, data has no payload,
// every other chunk its word-aligned payload
inline bool unpack_metadata(const uint8_t *m, size_t n, std::vector<WavChunk> &out) {
  size_t p = 0;
  while (p + 8 <= n) {
    const uint32_t id = rd32(m + p), sz = rd32(m + p + 4);
    p += 8;
    const size_t len = id == kIdRiff ? 4 : (id == kIdData ? 0 : word_align(sz));
    if (p + len > n) return false;
    out.push_back({id, sz, std::vector<uint8_t>(m + p, m + p + len)});
    p += len;
  }
  return p == n;
}
// Wav::WriteSamples (file/wav.cpp:124-160): planar int32 [ch][n] -> interleaved little-endian sample bytes
inline void pack_samples(const int32_t *pcm, long long ch_stride, int nch, int n, int bytes_per_sample, std::vector<uint8_t> &out) {
  for (int i = 0; i < n; i++)
    for (int k = 0; k < nch; k++) {
      const int32_t v = pcm[(size_t)k * ch_stride + i];
      if (bytes_per_sample == 1) out.push_back((uint8_t)((v + 128) & 0xff));
      else for (int b = 0; b < bytes_per_sample; b++) out.push_back((uint8_t)((uint32_t)v >> (8 * b)));
    }
}
// the WAV file Codec::DecodeFile writes (libsac.cpp:857-883): chunks up to and including the data header, the samples,
// a pad byte after an odd-sized data chunk, then the remaining chunks (Wav::WriteHeader, wav.cpp:265-281)
inline std::vector<uint8_t> rebuild_wav(const std::vector<WavChunk> &chunks, const std::vector<uint8_t> &data) {
  std::vector<uint8_t> o;
  size_t i = 0;
  while (i < chunks.size()) {
    const WavChunk &c = chunks[i++];
    wr32(o, c.id); wr32(o, c.size);
    if (c.id == kIdData) break;
    o.insert(o.end(), c.payload.begin(), c.payload.end());
  }
  o.insert(o.end(), data.begin(), data.end());
  if (data.size() & 1) o.push_back(0);
  for (; i < chunks.size(); i++) { wr32(o, chunks[i].id); wr32(o, chunks[i].size); o.insert(o.end(), chunks[i].payload.begin(), chunks[i].payload.end()); }
  return o;
}

struct SacFrameInfo { int numsamples; struct Ch { int blocksize, mean, minval, maxval, maxbpn, mapped; } ch[2]; };
// walks the frame records (WriteEncoded layout: u32 numsamples, 58 x f32 profile, per channel u32 blocksize, mean, min,
// max, u16 flag (bit 9 = mapped, low byte = maxbpn), payload); false on a truncated file
inline bool scan_sac_frames(const std::vector<uint8_t> &raw, const SacHeader &h, std::vector<SacFrameInfo> &out, long long *coef_hdr, long long *block_hdr) {
  size_t p = h.frames_at;
  *coef_hdr = *block_hdr = 0;
  while (p < raw.size()) {
    if (p + 4 + 58 * 4 > raw.size()) return false;
    SacFrameInfo f{}; f.numsamples = (int)rd32(&raw[p]); p += 4 + 58 * 4; *coef_hdr += 58 * 4;
    for (int c = 0; c < h.numchannels && c < 2; c++) {
      if (p + 18 > raw.size()) return false;
      const uint16_t flag = rd16(&raw[p + 16]);
      f.ch[c] = {(int)rd32(&raw[p]), (int)rd32(&raw[p + 4]), (int)rd32(&raw[p + 8]), (int)rd32(&raw[p + 12]), flag & 0xff, (flag >> 9) & 1};
      p += 18; *block_hdr += 18;
      if (p + (size_t)f.ch[c].blocksize > raw.size()) return false;
      p += (size_t)f.ch[c].blocksize;
    }
    out.push_back(f);
  }
  return true;
}

}  // namespace sacamd
