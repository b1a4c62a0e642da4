"""Round-5 measurements of what was only asserted before ("measure what is only asserted"):
  (a) decode throughput: N full-length stereo frames through sacamd_decode_frames (GPU: ReadEncoded + Decode + Unpredict of the
      whole batch) beside the genuine reference decoder (oracle/_ref, Codec::DecodeFile's frame loop, libsac.cpp:857-883) on one core;
  (b) the reference's DEFAULT search mode, OptDDS::run_single (num_threads = 0, opt/dds.cpp:33-60 -- what `--high` means without
      --opt-cfg): MSamples/s on a batch of full-length frames, and the record of frame 0 against the reference run the same way.
    python tests/gpu_measure_r5.py [--frames-dec 64] [--frames-single 256] > gpurun_out/r05/measure_r5.json   (one JSON line each)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sac_amd.api as api  # noqa: E402
from oracle_api import Checker, frame_cfg, ref_available  # noqa: E402
from sac_amd.synth import synth_pcm  # noqa: E402

RATE = 44100
N = 20 * RATE


def frames_of(k, seed0):
    return [synth_pcm(N, 2, seed=seed0 + i, rate=RATE) for i in range(k)]


def decode_throughput(k):
    frames = frames_of(k, 1000)                       # the bench's own frames 0..k-1
    ctx = api.Context(2, N, k)
    ctx.upload_i32(frames, N)
    cfg = api.make_cfg("normal")
    ctx.analyse(cfg)
    recs, _ = ctx.encode_frames(cfg)
    t = time.time()
    dec, _ = ctx.decode_frames(recs, N)
    dt = time.time() - t
    ok = all(np.array_equal(d, f) for d, f in zip(dec, frames))
    ctx.close()
    out = {"measure": "decode", "frames": k, "frame_seconds": 20, "gpu_seconds": dt, "gpu_MSamples_s": k * 2 * N / dt / 1e6, "gpu_x_realtime": k * 20 / dt,
           "lossless": bool(ok), "records": "--normal (default profile) records of the bench's frames 0..k-1; decode cost does not depend on how the profile was found"}
    kind = "ref" if ref_available() else "orc"
    chk = Checker(kind)
    t = time.time()
    d0, _ = chk.decode_frame(recs[0], 2, N)
    dc = time.time() - t
    out["cpu_baseline"] = {"kind": "reference" if kind == "ref" else "port", "cores": 1, "seconds_frame0": dc, "MSamples_s": 2 * N / dc / 1e6, "x_realtime": 20 / dc,
                           "same_pcm": bool(np.array_equal(d0, frames[0])), "sample": "frame 0 of the batch, FrameCoder::ReadEncoded + Decode + Unpredict (libsac.cpp:857-883), 1 core"}
    out["speedup_vs_one_core"] = out["gpu_MSamples_s"] / out["cpu_baseline"]["MSamples_s"]
    return out


def single_search(k):
    frames = frames_of(k, 1000)
    ctx = api.Context(2, N, k)
    ctx.upload_i32(frames, N)
    cfg = api.make_cfg("high", num_threads=0, reset=1)          # --high --opt-reset: OptDDS::run_single, 100 sequential evaluations
    t = time.time()
    ctx.analyse(cfg)
    recs, prof = ctx.encode_frames(cfg)
    dt = time.time() - t
    kt = ctx.kernel_times()
    ctx.close()
    out = {"measure": "run_single", "frames": k, "frame_seconds": 20, "config": "--high --opt-reset (num_threads = 0: OptDDS::run_single, the reference's default)",
           "gpu_seconds": dt, "gpu_MSamples_s": k * 2 * N / dt / 1e6, "bps": 8 * sum(len(r) for r in recs) / (k * 2 * N),
           "kernel_ms": {a: round(v["ms"], 1) for a, v in kt.items()}}
    kind = "ref" if ref_available() else "orc"
    chk = Checker(kind)
    t = time.time()
    r = chk.encode_frame(frames[0], frame_cfg("high", num_threads=0, reset=1), N)
    dc = time.time() - t
    out["cpu_baseline"] = {"kind": "reference" if kind == "ref" else "port", "cores": 1, "seconds_frame0": dc, "MSamples_s": 2 * N / dc / 1e6,
                           "sample": "frame 0, the same configuration, 1 core"}
    out["record0_equals_reference"] = bool(bytes(recs[0]) == bytes(r["record"]))
    out["speedup_vs_one_core"] = out["gpu_MSamples_s"] / out["cpu_baseline"]["MSamples_s"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames-dec", type=int, default=64)
    ap.add_argument("--frames-single", type=int, default=256)
    a = ap.parse_args()
    if a.frames_dec > 0:
        print(json.dumps(decode_throughput(a.frames_dec)), flush=True)
    if a.frames_single > 0:
        print(json.dumps(single_search(a.frames_single)), flush=True)


if __name__ == "__main__":
    main()
