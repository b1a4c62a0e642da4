"""Encode WAV files to .sac with the MI355X frame encoder.

    python -m sac_amd.cli encode in.wav out.sac [--mode high] [--dds-n 8] [--framelen 20] [--no-adapt-block]
    python -m sac_amd.cli list file.sac
    python -m sac_amd.cli decode file.sac out.wav      (GPU decoder: sacamd_decode_frames over all frame records at once)

The encode side of the reference's command line (cmdline.cpp): presets --normal .. --insane,
--opt-cfg=dds,N (here --dds-n), --framelen, --adapt-block.  Frames are always independent
(--opt-reset), which is what makes them batchable.  `decode` is the reference's --decode (cmdline.cpp:295-358,
Codec::DecodeFile libsac.cpp:857-883) with all frames of the file decoded as one GPU batch; the files are
byte-compatible with the reference's in both directions.
"""
import argparse
import sys

from . import api, container


def main(argv=None):
    ap = argparse.ArgumentParser(prog="sac_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    e = sub.add_parser("encode")
    e.add_argument("wav"); e.add_argument("sac")
    e.add_argument("--mode", default="normal", choices=["normal", "high", "veryhigh", "extrahigh", "best", "insane"])
    e.add_argument("--dds-n", type=int, default=8)
    e.add_argument("--framelen", type=int, default=20)
    e.add_argument("--no-adapt-block", action="store_true")
    e.add_argument("--max-frames", type=int, default=256, help="frames per GPU batch")
    ls = sub.add_parser("list")
    ls.add_argument("sac")
    d = sub.add_parser("decode")
    d.add_argument("sac"); d.add_argument("wav")
    a = ap.parse_args(argv)
    if a.cmd == "decode":
        import hashlib
        hdr, md5, chunks, recs = container.read_sac(a.sac)
        nch, rate, bits, nsamp, framelen = hdr["numchannels"], hdr["samplerate"], hdr["bitspersample"], hdr["numsamples"], hdr["max_framelen"]
        data = b""
        if recs:
            ctx = api.Context(nch, framelen * rate, min(len(recs), 256))
            for b0 in range(0, len(recs), 256):
                pcm, _ = ctx.decode_frames(recs[b0: b0 + 256], framelen * rate)
                data += b"".join(container.sample_bytes(p, bits) for p in pcm)
            ctx.close()
        ok = hashlib.md5(data).digest() == md5
        open(a.wav, "wb").write(container.rebuild_wav(chunks, data))
        print(f"{a.sac}: {len(recs)} frames, {len(data)} sample bytes -> {a.wav}; Audio MD5: {'ok' if ok else 'Error'}")
        return 0 if ok else 1
    if a.cmd == "list":
        hdr, md5, chunks, recs = container.read_sac(a.sac)
        print(hdr, "md5", md5.hex(), "frames", len(recs), "chunks", [(hex(c), s) for c, s, _ in chunks])
        return 0
    blob = open(a.wav, "rb").read()
    w = container.parse_wav(blob)
    ctx = api.Context(w.numchannels, a.framelen * w.samplerate, a.max_frames)
    cfg = api.make_cfg(a.mode, num_threads=a.dds_n if a.mode != "normal" else 0, reset=1)
    (info, recs), = container.encode_wav_files(ctx, [blob], cfg, max_framelen=a.framelen, adapt_block=not a.no_adapt_block)
    ctx.close()
    size = container.write_sac(a.sac, info, a.framelen, recs)
    print(f"{a.wav}: {info.numsamples} samples x {info.numchannels} ch -> {size} bytes, "
          f"{8 * size / max(1, info.numsamples * info.numchannels):.3f} bps, {len(recs)} frames")
    return 0


if __name__ == "__main__":
    sys.exit(main())
