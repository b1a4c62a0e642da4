// sac_amd/csrc/sacenc.cpp -- command-line encoder on top of the C ABI: WAV -> .sac on an MI355X.
//
//   sacenc [--normal|--high|--veryhigh|--extrahigh|--best|--insane] [--opt-cfg=dds,N] [--framelen=S]
//          [--adapt-block=no] [--max-frames=N] in.wav [more.wav ...] out.sac|outdir
//   sacenc --list|--listfull file.sac       header, ratio, MD5 (and every frame record) as the reference's --list / --listfull
//   sacenc --decode file.sac out.wav        .sac -> WAV on the GPU (cmdline.cpp:295-358, Codec::DecodeFile libsac.cpp:857-883):
//                                           all frame records of the file decoded as one batch (sacamd_decode_frames),
//                                           MD5 of the sample bytes checked against the header's
//
//   sacenc ... --world=W --rank=R --comm-id=FILE [--device=D] in.wav ... out        one process per GPU (the C++ host side of
//                                           the multi-GPU path): every rank plans the same frame list, sacamd_assign_frames
//                                           splits it by cost, each rank encodes its share on its GPU, sacamd_gather_records
//                                           (RCCL) brings the records to rank 0 in frame order, rank 0 writes the files.  FILE
//                                           carries the RCCL unique id from rank 0 to the others (any shared path).
//
// The encode side of the reference's command line (/root/reference/src/cmdline.cpp:127-235) and of
// Codec::EncodeFile (libsac/libsac.cpp:782-855): reads of framelen seconds, adaptive sub-frame split
// (sacamd_plan_subframes), every frame of every input file staged as ONE batch per max-frames
// (frames are independent: --opt-reset semantics), records written behind the SAC2 header + MD5.
// With several inputs the last argument is a directory.  Host code only; no CPU compute path.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>
#include <unistd.h>

#include "../../include/sac_amd.h"
#include "sacfile.h"

using namespace sacamd;

static std::vector<uint8_t> slurp(const std::string &p) {
  std::ifstream f(p, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + p);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct FrameRef { int file, start, length; };

// communicator of a multi-rank run, reachable from the error path: a rank that fails before the gather joins it with
// nrec = -1, so that the other ranks return from sacamd_gather_records with an error instead of waiting for it
static sacamd_comm *g_comm = nullptr;
static bool g_gather_entered = false;

int main(int argc, char **argv) {
  try {
    sacamd_cfg cfg; sacamd_default_cfg(&cfg);
    cfg.reset = 1;
    int framelen = 20, adapt_block = 1, max_frames = 256;
    int list_mode = 0;             // 1: --list, 2: --listfull (host only)
    bool decode_mode = false;      // --decode
    int world = 1, rank = 0, device = -1;   // --world / --rank / --device: one process per GPU
    std::string comm_id_file;
    bool job_tag_given = false;
    unsigned long long job_tag = 0; // --job-tag=N: ties the rendezvous file to this job (stale files of other runs are ignored)
    bool force_gather = false;     // --force-gather: take the communicator / gather path with one rank too (one-GPU test)
    bool header_only = false;      // --header-only: write header + MD5 of each input and stop (no device needed; tests)
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
      std::string a = argv[i];
      auto preset = [&](int opt, double frac, int e, double sig, int cost) { cfg.optimize = opt; cfg.fraction = frac; cfg.maxnfunc = e; cfg.sigma = sig; cfg.optimize_cost = cost; };
      if (a == "--normal") preset(0, 0.0, 0, 0.2, SACAMD_COST_ENTROPY);                 // cmdline.cpp:127-156
      else if (a == "--high") preset(1, 0.1, 100, 0.20, SACAMD_COST_ENTROPY);
      else if (a == "--veryhigh") preset(1, 0.2, 300, 0.25, SACAMD_COST_ENTROPY);
      else if (a == "--extrahigh") preset(1, 0.2, 600, 0.25, SACAMD_COST_ENTROPY);
      else if (a == "--best") preset(1, 0.5, 1000, 0.25, SACAMD_COST_BITPLANE);
      else if (a == "--insane") preset(1, 0.5, 1500, 0.25, SACAMD_COST_BITPLANE);
      else if (a.rfind("--opt-cfg=", 0) == 0) {                                           // cmdline.cpp:195-207: method[,threads[,sigma]]
        std::string v = a.substr(10), m = v.substr(0, v.find(','));
        for (auto &ch : m) ch = (char)std::toupper((unsigned char)ch);
        if (m == "DDS") cfg.optimize_search = SACAMD_SEARCH_DDS; else if (m == "DE") cfg.optimize_search = SACAMD_SEARCH_DE;
        else if (m == "CMA") cfg.optimize_search = SACAMD_SEARCH_CMA; else std::cerr << "  warning: invalid opt='" << m << "'\n";
        size_t c1 = v.find(',');
        if (c1 != std::string::npos) {
          cfg.num_threads = std::min(std::max(std::atoi(v.c_str() + c1 + 1), 0), 256);
          size_t c2 = v.find(',', c1 + 1);
          if (c2 != std::string::npos) cfg.sigma = std::min(std::max(std::atof(v.c_str() + c2 + 1), 0.0), 1.0);
        }
      }
      else if (a == "--opt-reset") cfg.reset = 1;                                         // cmdline.cpp:193-194 (the batch encoder's default: frames are independent)
      else if (a.rfind("--framelen=", 0) == 0) framelen = std::atoi(a.c_str() + 11);
      else if (a == "--adapt-block=no" || a == "--adapt-block=0") adapt_block = 0;
      else if (a == "--sparse-pcm=no" || a == "--sparse-pcm=0") cfg.sparse_pcm = 0;                         // cmdline.cpp:187-190
      else if (a == "--sparse-pcm" || a == "--sparse-pcm=yes" || a == "--sparse-pcm=1") cfg.sparse_pcm = 1;
      else if (a.rfind("--max-frames=", 0) == 0) max_frames = std::atoi(a.c_str() + 13);
      else if (a == "--header-only") header_only = true;
      else if (a == "--list") list_mode = 1;
      else if (a == "--listfull") list_mode = 2;
      else if (a == "--decode") decode_mode = true;
      else if (a.rfind("--world=", 0) == 0) world = std::atoi(a.c_str() + 8);
      else if (a.rfind("--rank=", 0) == 0) rank = std::atoi(a.c_str() + 7);
      else if (a.rfind("--device=", 0) == 0) device = std::atoi(a.c_str() + 9);
      else if (a.rfind("--comm-id=", 0) == 0) comm_id_file = a.substr(10);
      else if (a.rfind("--job-tag=", 0) == 0) { job_tag = std::strtoull(a.c_str() + 10, nullptr, 10); job_tag_given = true; }
      else if (a == "--force-gather") force_gather = true;
      else if (a.rfind("--", 0) == 0) { std::cerr << "unknown option " << a << "\n"; return 2; }
      else pos.push_back(a);
    }
    if (list_mode) {               // cmdline.cpp:295-323 + Codec::ScanFrames (libsac.cpp:659-693)
      if (pos.size() != 1) { std::cerr << "usage: sacenc --list|--listfull file.sac\n"; return 2; }
      const std::vector<uint8_t> raw = slurp(pos[0]);
      SacHeader h;
      if (!read_sac_header(raw, h)) { std::cout << "warning: input is not a valid .sac file\n"; return 1; }
      const double bps = (double)raw.size() * 8.0 / ((double)h.numsamples * h.numchannels);
      std::printf("Open: '%s': ok (%zu Bytes)\n", pos[0].c_str(), raw.size());
      std::printf("  WAVE  Codec: PCM (%d kbps)\n", (int)std::lround(h.samplerate * h.numchannels * bps / 1000.0));
      std::printf("  %dHz %d Bit  %d channel(s)  %d samples  metadata %d bytes\n", h.samplerate, h.bitspersample, h.numchannels, h.numsamples, h.metadatasize);
      std::printf("  Profile: %ds\n  Ratio:   %.3f bps\n\n  Audio MD5: ", h.max_framelen, bps);
      for (int i = 0; i < 16; i++) std::printf("%x", h.md5[i]);
      std::printf("\n");
      if (list_mode == 2) {
        std::vector<SacFrameInfo> fr; long long ch = 0, bh = 0;
        const bool ok = scan_sac_frames(raw, h, fr, &ch, &bh);
        for (size_t i = 0; i < fr.size(); i++) {
          std::printf("Frame %zu: %d samples \n", i + 1, fr[i].numsamples);
          for (int c = 0; c < h.numchannels && c < 2; c++) {
            std::printf("  Channel %d: %d bytes\n    Bpn: %d, sparse_pcm: %d\n    mean: %d, min: %d, max: %d\n", c, fr[i].ch[c].blocksize, fr[i].ch[c].maxbpn,
                        fr[i].ch[c].mapped, fr[i].ch[c].mean, fr[i].ch[c].minval, fr[i].ch[c].maxval);
          }
        }
        std::printf("Frames   %zu\nHdr_size %lld (coefs %lld,block %lld)\n", fr.size(), ch + bh, ch, bh);
        if (!ok) { std::printf("warning: truncated frame record\n"); return 1; }
      }
      return 0;
    }
    if (decode_mode) {
      if (pos.size() != 2) { std::cerr << "usage: sacenc --decode file.sac out.wav\n"; return 2; }
      const std::vector<uint8_t> raw = slurp(pos[0]);
      SacHeader h;
      if (!read_sac_header(raw, h)) { std::cout << "warning: input is not a valid .sac file\n"; return 1; }
      std::vector<WavChunk> chunks;
      if (!unpack_metadata(raw.data() + 22, (size_t)h.metadatasize, chunks)) std::cerr << "  warning: unpackmetadata mismatch\n";
      std::vector<SacFrameInfo> fr; long long chdr = 0, bhdr = 0;
      if (!scan_sac_frames(raw, h, fr, &chdr, &bhdr)) throw std::runtime_error("truncated .sac file");
      // record offsets (frame f = [off[f], off[f+1]) behind the header)
      std::vector<long long> off(1, 0);
      for (auto &f : fr) { long long len = 4 + 58 * 4; for (int c = 0; c < h.numchannels && c < 2; c++) len += 18 + f.ch[c].blocksize; off.push_back(off.back() + len); }
      const int maxfs = h.samplerate * h.max_framelen, nf = (int)fr.size();
      int bps_bytes = (h.bitspersample + 7) / 8;
      for (auto &c : chunks) if (c.id == kIdFmt && c.payload.size() >= 14 && h.numchannels > 0) bps_bytes = rd16(&c.payload[12]) / h.numchannels;   // blockalign / channels
      std::vector<uint8_t> data;
      data.reserve((size_t)h.numsamples * h.numchannels * bps_bytes);
      if (nf > 0) {
        sacamd_ctx *ctx = nullptr;
        const int batch = std::min(nf, max_frames);
        if (sacamd_ctx_create(0, h.numchannels, maxfs, batch, &ctx) != 0) throw std::runtime_error("no usable gfx950 device (sacamd_ctx_create failed)");
        std::vector<int32_t> pcm((size_t)batch * h.numchannels * maxfs);
        std::vector<int> ns(batch);
        for (int b0 = 0; b0 < nf; b0 += batch) {
          const int nb = std::min(batch, nf - b0);
          std::vector<long long> o(nb + 1);
          for (int i = 0; i <= nb; i++) o[i] = off[b0 + i] - off[b0];
          if (sacamd_decode_frames(ctx, nb, maxfs, raw.data() + h.frames_at + off[b0], o.data(), pcm.data(), (long long)h.numchannels * maxfs, maxfs, ns.data(), nullptr) != 0)
            throw std::runtime_error(std::string("sac_amd: ") + sacamd_last_error(ctx));
          for (int i = 0; i < nb; i++) pack_samples(&pcm[(size_t)i * h.numchannels * maxfs], maxfs, h.numchannels, ns[i], bps_bytes, data);
        }
        sacamd_ctx_destroy(ctx);
      }
      Md5 md; md.update(data.data(), data.size());
      uint8_t dig[16]; md.finish(dig);
      const std::vector<uint8_t> wav = rebuild_wav(chunks, data);
      std::ofstream o(pos[1], std::ios::binary);
      if (!o) { std::cout << "could not create\n"; return 1; }
      o.write((const char *)wav.data(), (std::streamsize)wav.size());
      const bool md5ok = std::memcmp(dig, h.md5, 16) == 0;
      std::printf("%s: %d frames, %zu sample bytes -> %s\n  Audio MD5: %s\n", pos[0].c_str(), nf, data.size(), pos[1].c_str(), md5ok ? "ok" : "Error");
      return md5ok ? 0 : 1;
    }
    if (pos.size() < 2 || framelen < 1 || framelen > 255 || max_frames < 1) { std::cerr << "usage: sacenc [options] in.wav [more.wav ...] out.sac|outdir\n"; return 2; }
    const std::string outarg = pos.back(); pos.pop_back();
    const bool multi = pos.size() > 1;

    std::vector<WavInfo> wavs;
    std::vector<std::vector<int32_t>> pcm;
    for (auto &p : pos) { wavs.push_back(parse_wav(slurp(p))); pcm.push_back(pcm_from_wav(wavs.back())); }
    const int nch = wavs[0].numchannels, rate = wavs[0].samplerate;
    for (auto &w : wavs) if (w.numchannels != nch || w.samplerate != rate) throw std::runtime_error("all inputs of one run must share channel count and sample rate");
    const int maxfs = framelen * rate;
    if (header_only) {
      if (multi) throw std::runtime_error("--header-only takes one input");
      const std::vector<uint8_t> hdr = sac_header_and_md5(wavs[0], framelen);
      std::ofstream o(outarg, std::ios::binary);
      o.write((const char *)hdr.data(), (std::streamsize)hdr.size());
      return 0;
    }

    const bool use_comm = world > 1 || force_gather;
    if (world < 1 || rank < 0 || rank >= world || (use_comm && comm_id_file.empty())) { std::cerr << "sacenc: --world=W --rank=R (0 <= R < W) --comm-id=FILE\n"; return 2; }
    if (device < 0) device = rank;                     // one process per GPU: rank r drives GPU r unless told otherwise
    if (world > 1 && !job_tag_given) {
      // no --job-tag: take what the launcher provides (the same value on every rank of ONE job); none -> refuse, because with a
      // shared default tag a rank could pick up the id file a crashed run of the same world size left behind and hang in
      // ncclCommInitRank (round-4 advice)
      for (const char *var : {"SACENC_JOB_TAG", "SLURM_JOB_ID", "TORCHELASTIC_RUN_ID"}) {   // (not MASTER_PORT: the same for consecutive runs, no protection against a stale file)
        const char *e = std::getenv(var);
        if (e && *e) { unsigned long long h = 1469598103934665603ull; for (const char *q = e; *q; q++) h = (h ^ (unsigned char)*q) * 1099511628211ull; job_tag = h; job_tag_given = true; break; }
      }
      if (!job_tag_given) { std::cerr << "sacenc: --world > 1 needs --job-tag=N (or SACENC_JOB_TAG / SLURM_JOB_ID / TORCHELASTIC_RUN_ID in the environment), the same on every rank of this job\n"; return 2; }
    }
    if (use_comm && !cfg.reset) throw std::runtime_error("frames are sharded across ranks: --opt-reset semantics only");
    sacamd_ctx *ctx = nullptr;
    if (sacamd_ctx_create(device, nch, maxfs, max_frames, &ctx) != 0) throw std::runtime_error("no usable gfx950 device (sacamd_ctx_create failed)");
    auto chk = [&](int rc) { if (rc != 0) throw std::runtime_error(std::string("sac_amd: ") + sacamd_last_error(ctx)); };
    sacamd_comm *comm = nullptr;
    if (use_comm) {
      // RCCL communicator; the unique id travels through FILE = magic, world, job tag, id.  Rank 0 removes whatever a previous
      // run left there before it writes (rename: readers never see a partial file) and again once every rank has joined;
      // the other ranks take a file only when magic, world size and --job-tag agree -- give every job its own tag (launcher
      // PID, time stamp) and a stale file of a crashed run can never be mistaken for this job's (round-3 advice).
      struct IdFile { char magic[8]; uint32_t world; uint32_t pad; uint64_t tag; uint8_t id[SACAMD_COMM_ID_BYTES]; } idf;
      static_assert(sizeof(IdFile) == 24 + SACAMD_COMM_ID_BYTES, "id file layout");
      if (rank == 0) {
        (void)std::remove(comm_id_file.c_str());
        std::memset(&idf, 0, sizeof(idf));
        std::memcpy(idf.magic, "SACAMDID", 8); idf.world = (uint32_t)world; idf.tag = job_tag;
        if (sacamd_comm_unique_id(idf.id) != 0) throw std::runtime_error("sacamd_comm_unique_id failed");
        { std::ofstream o(comm_id_file + ".tmp", std::ios::binary); o.write((const char *)&idf, sizeof(idf)); }
        if (std::rename((comm_id_file + ".tmp").c_str(), comm_id_file.c_str()) != 0) throw std::runtime_error("cannot write " + comm_id_file);
      } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; tries++) {             // up to 10 minutes
          std::ifstream f(comm_id_file, std::ios::binary);
          if (f && f.read((char *)&idf, sizeof(idf)) && std::memcmp(idf.magic, "SACAMDID", 8) == 0 && idf.world == (uint32_t)world && idf.tag == job_tag) got = true;
          else usleep(100000);
        }
        if (!got) throw std::runtime_error("no RCCL unique id for this job (world " + std::to_string(world) + ", tag " + std::to_string(job_tag) + ") in " + comm_id_file);
      }
      if (sacamd_comm_create(device, rank, world, idf.id, &comm) != 0) throw std::runtime_error("sacamd_comm_create failed");
      g_comm = comm;
      if (rank == 0) (void)std::remove(comm_id_file.c_str());            // ncclCommInitRank returned: every rank has read the id
    }

    // frame list: reads of maxfs samples, each cut into sub-frames (libsac.cpp:805-820)
    std::vector<FrameRef> frames;
    for (size_t f = 0; f < wavs.size(); f++) {
      const int total = wavs[f].numsamples;
      for (int p0 = 0; p0 < total; p0 += maxfs) {
        const int n = std::min(maxfs, total - p0);
        if (adapt_block) {
          sacamd_subframe sf[64]; int cnt = 0;
          chk(sacamd_plan_subframes(ctx, pcm[f].data() + p0, total, nch, n, 3 * rate, 3 * rate, sf, 64, &cnt));
          for (int i = 0; i < cnt; i++) frames.push_back({(int)f, p0 + sf[i].start, sf[i].length});
        } else frames.push_back({(int)f, p0, n});
      }
    }

    // multi-GPU: this rank's share of the frame list (longest first by estimated cost channels * (E * T_opt + T));
    // every rank computes the same assignment
    const std::vector<FrameRef> all_frames = frames;
    std::vector<int> my_ids(all_frames.size());
    for (size_t i = 0; i < all_frames.size(); i++) my_ids[i] = (int)i;
    if (use_comm) {
      std::vector<double> cost(all_frames.size());
      std::vector<int> owner(all_frames.size());
      for (size_t i = 0; i < all_frames.size(); i++) {
        const double T = all_frames[i].length, Topt = std::min(T, std::ceil(maxfs * cfg.fraction));
        cost[i] = nch * ((cfg.optimize ? cfg.maxnfunc : 0) * Topt + T);
      }
      chk(sacamd_assign_frames(cost.data(), (int)cost.size(), world, owner.data()));
      frames.clear(); my_ids.clear();
      for (size_t i = 0; i < all_frames.size(); i++) if (owner[i] == rank) { frames.push_back(all_frames[i]); my_ids.push_back((int)i); }
    }

    // encode in batches of max_frames
    std::vector<std::vector<uint8_t>> payload(wavs.size());
    std::vector<int> nfr(wavs.size(), 0);
    std::vector<uint8_t> my_recs;                    // multi-GPU: this rank's records back to back, my_off[i]..my_off[i+1]
    std::vector<long long> my_off(1, 0);
    for (size_t b0 = 0; b0 < frames.size(); b0 += (size_t)max_frames) {
      const int nb = (int)std::min((size_t)max_frames, frames.size() - b0);
      int stride = 0;
      for (int i = 0; i < nb; i++) stride = std::max(stride, frames[b0 + i].length);
      std::vector<int32_t> stage((size_t)nb * nch * stride, 0);
      std::vector<int> ns(nb);
      long long cap = 0;
      for (int i = 0; i < nb; i++) {
        const FrameRef &fr = frames[b0 + i];
        ns[i] = fr.length;
        for (int ch = 0; ch < nch; ch++)
          std::memcpy(&stage[((size_t)i * nch + ch) * stride], pcm[fr.file].data() + (size_t)ch * wavs[fr.file].numsamples + fr.start, sizeof(int32_t) * (size_t)fr.length);
        cap += (long long)fr.length * nch * 4 + 2 * 4096 + 70000;
      }
      chk(sacamd_frames_upload_i32(ctx, nb, maxfs, stage.data(), (long long)nch * stride, stride, ns.data()));
      std::vector<float> prof((size_t)nb * SACAMD_NUM_COEFS), vmin(SACAMD_NUM_COEFS), vmax(SACAMD_NUM_COEFS), vdef(SACAMD_NUM_COEFS);
      sacamd_default_profile(vmin.data(), vmax.data(), vdef.data());
      for (int i = 0; i < nb; i++) std::copy(vdef.begin(), vdef.end(), prof.begin() + (size_t)i * SACAMD_NUM_COEFS);
      std::vector<uint8_t> out((size_t)cap);
      std::vector<long long> off(nb + 1);
      chk(sacamd_encode_frames(ctx, &cfg, prof.data(), out.data(), cap, off.data()));
      for (int i = 0; i < nb; i++) {
        if (use_comm) { my_recs.insert(my_recs.end(), out.begin() + off[i], out.begin() + off[i + 1]); my_off.push_back((long long)my_recs.size()); continue; }
        auto &dst = payload[frames[b0 + i].file];
        dst.insert(dst.end(), out.begin() + off[i], out.begin() + off[i + 1]);
        nfr[frames[b0 + i].file]++;
      }
    }
    sacamd_ctx_destroy(ctx);
    if (use_comm) {
      // the one exchange of the path: all records to rank 0 in frame order (== the order WriteEncoded appends them)
      const int total = (int)all_frames.size();
      long long cap = 0;
      for (auto &fr : all_frames) cap += (long long)fr.length * nch * 4 + 2 * 4096 + 70000;
      std::vector<uint8_t> all(rank == 0 ? (size_t)cap : 0);
      std::vector<long long> all_off(rank == 0 ? (size_t)total + 1 : 0);
      if (my_recs.empty()) my_recs.push_back(0);
      g_gather_entered = true;
      const int rc = sacamd_gather_records(comm, (int)my_ids.size(), my_ids.data(), my_recs.data(), my_off.data(), total,
                                           rank == 0 ? all.data() : nullptr, rank == 0 ? cap : 0, rank == 0 ? all_off.data() : nullptr);
      if (rc != 0) throw std::runtime_error(std::string("sacamd_gather_records: ") + sacamd_comm_last_error(comm));
      sacamd_comm_destroy(comm); g_comm = nullptr;
      if (rank != 0) return 0;
      for (int i = 0; i < total; i++) {
        auto &dst = payload[all_frames[i].file];
        dst.insert(dst.end(), all.begin() + all_off[i], all.begin() + all_off[i + 1]);
        nfr[all_frames[i].file]++;
      }
    }

    for (size_t f = 0; f < wavs.size(); f++) {
      std::string op = outarg;
      if (multi) {
        std::string base = pos[f].substr(pos[f].find_last_of('/') == std::string::npos ? 0 : pos[f].find_last_of('/') + 1);
        const size_t dot = base.find_last_of('.');
        op = outarg + "/" + (dot == std::string::npos ? base : base.substr(0, dot)) + ".sac";
      }
      const std::vector<uint8_t> hdr = sac_header_and_md5(wavs[f], framelen);
      std::ofstream o(op, std::ios::binary);
      if (!o) throw std::runtime_error("cannot write " + op);
      o.write((const char *)hdr.data(), (std::streamsize)hdr.size());
      o.write((const char *)payload[f].data(), (std::streamsize)payload[f].size());
      const double total = (double)hdr.size() + payload[f].size();
      std::printf("%s: %d samples x %d ch, %d frames -> %.0f bytes, %.3f bps\n", pos[f].c_str(), wavs[f].numsamples, nch, nfr[f], total,
                  8.0 * total / std::max(1.0, (double)wavs[f].numsamples * nch));
    }
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "sacenc: " << e.what() << "\n";
    if (g_comm) {                  // tell the other ranks (they are in, or on their way to, the gather), then drop the communicator
      if (!g_gather_entered) (void)sacamd_gather_records(g_comm, -1, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr);
      sacamd_comm_destroy(g_comm);
    }
    return 1;
  }
}
