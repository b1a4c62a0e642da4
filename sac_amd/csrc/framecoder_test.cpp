// sac_amd/csrc/framecoder_test.cpp -- drives sacamd::FrameCoder (framecoder.h) exactly as the reference's
// Codec::EncodeFile drives its FrameCoder (/root/reference/src/libsac/libsac.cpp:788,822-829): fill samples[ch],
// SetNumSamples, Predict(), Encode(), WriteEncoded().  Test program (tests/test_gpu_parity.py compares the record it
// writes with the golden record of the genuine reference).
//   framecoder_test <in.i32 planar [nframes][nch][n]> <nch> <n> <framesize> <optimize> <fraction> <maxnfunc> <num_threads> <sigma> <out.rec> [reset=1 [nframes=1]]
// With nframes > 1 the ONE FrameCoder encodes the frames one after the other and writes their records back to back, so that
// with reset=0 (the reference's default) each search starts from the previous frame's optimum (libsac.cpp:461-466).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include "framecoder.h"

// what the reference hands to WriteEncoded / ReadEncoded: an object with a `file` stream (AudioFile, file/file.h:10-36)
struct AudioFileLike { std::fstream file; };

// --decode: Codec::DecodeFile's frame loop (libsac.cpp:857-883): ReadEncoded, Decode, Unpredict, then the samples of every channel
//   framecoder_test --decode <in.rec> <nch> <framesize> <nframes> <out.i32 planar, frame after frame [nch][numsamples]>
static int decode_main(int argc, char **argv) {
  if (argc != 7) { std::fprintf(stderr, "usage: framecoder_test --decode in.rec nch framesize nframes out.i32\n"); return 2; }
  const int nch = std::atoi(argv[3]), framesize = std::atoi(argv[4]), nframes = std::atoi(argv[5]);
  try {
    sacamd::FrameCoder::tsac_cfg cfg;
    sacamd::FrameCoder coder(nch, framesize, cfg);
    AudioFileLike in; in.file.open(argv[2], std::ios::in | std::ios::binary);
    std::ofstream out(argv[6], std::ios::binary);
    if (!in.file || !out) { std::fprintf(stderr, "cannot open files\n"); return 2; }
    for (int f = 0; f < nframes; f++) {
      coder.ReadEncoded(in);
      coder.Decode();
      coder.Unpredict();
      for (int ch = 0; ch < nch; ch++) out.write(reinterpret_cast<const char *>(coder.samples[ch].data()), sizeof(int32_t) * (size_t)coder.GetNumSamples());
    }
  } catch (const std::exception &e) { std::fprintf(stderr, "framecoder_test: %s\n", e.what()); return 1; }
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::string(argv[1]) == "--decode") return decode_main(argc, argv);
  if (argc < 11 || argc > 13) { std::fprintf(stderr, "usage: framecoder_test in.i32 nch n framesize optimize fraction maxnfunc num_threads sigma out.rec [reset [nframes]]\n"); return 2; }
  const int nch = std::atoi(argv[2]), n = std::atoi(argv[3]), framesize = std::atoi(argv[4]);
  sacamd::FrameCoder::tsac_cfg cfg;
  cfg.optimize = std::atoi(argv[5]);
  cfg.ocfg.fraction = std::atof(argv[6]); cfg.ocfg.maxnfunc = std::atoi(argv[7]); cfg.ocfg.num_threads = std::atoi(argv[8]);
  cfg.ocfg.sigma = std::atof(argv[9]); cfg.ocfg.reset = argc > 11 ? std::atoi(argv[11]) : 1;
  const int nframes = argc > 12 ? std::atoi(argv[12]) : 1;
  try {
    sacamd::FrameCoder coder(nch, framesize, cfg);
    std::ifstream in(argv[1], std::ios::binary);
    AudioFileLike out; out.file.open(argv[10], std::ios::out | std::ios::binary | std::ios::trunc);
    for (int f = 0; f < nframes; f++) {
      for (int ch = 0; ch < nch; ch++) in.read(reinterpret_cast<char *>(coder.samples[ch].data()), sizeof(int32_t) * (size_t)n);
      if (!in) { std::fprintf(stderr, "short input\n"); return 2; }
      coder.SetNumSamples(n);
      coder.Predict();
      coder.Encode();
      coder.WriteEncoded(out);
    }
  } catch (const std::exception &e) { std::fprintf(stderr, "framecoder_test: %s\n", e.what()); return 1; }
  return 0;
}
