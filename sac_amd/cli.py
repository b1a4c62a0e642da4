"""Encode WAV files to .sac with the MI355X frame encoder.

    python -m sac_amd.cli encode in.wav out.sac [--mode high] [--dds-n 8] [--framelen 20] [--no-adapt-block]
    python -m sac_amd.cli list file.sac

The encode side of the reference's command line (cmdline.cpp): presets --normal .. --insane,
--opt-cfg=dds,N (here --dds-n), --framelen, --adapt-block.  Frames are always independent
(--opt-reset), which is what makes them batchable.  Decoding is done by the reference decoder:
the files are byte-compatible.
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
    a = ap.parse_args(argv)
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
