"""BASELINE configs[3] (--best) and configs[4] (--veryhigh, mixed 8-bit mono + 16-bit stereo) timed once at their real sizes
(full 20-s frames, real search windows and cost functions) with the evaluation count cut to what fits a few GPU minutes,
and one record of each compared with the genuine reference's (tests/golden/ref_golden_r5.npz: SHA-256 + length + profile).

    python tests/gpu_baseline_configs.py [--frames-best 16] [--frames-vh 32] > gpurun_out/r03/configs34.json

Prints one JSON line per configuration.  Frame 0 of every batch is the golden case; the other frames are further seeds."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sac_amd.api as api  # noqa: E402
from golden_cases import FULL_FRAMESIZE, FULL_RATE, config34_cases  # noqa: E402
from sac_amd.synth import synth_pcm  # noqa: E402


def gpu_cfg(cfg):
    return api.Cfg(cfg.optimize, cfg.sparse_pcm, cfg.zero_mean, cfg.reset, cfg.fraction, cfg.maxnfunc, cfg.num_threads, cfg.sigma, cfg.optk, cfg.cost, 0)


def run(name, nframes, nch, bits, seed0, golden, cases=None):
    raw0, cfg = (cases or config34_cases())[name]
    frames = [raw0] + [synth_pcm(20 * FULL_RATE, nch, seed0 + i, FULL_RATE, bits=bits) for i in range(1, nframes)]
    ctx = api.Context(nch, FULL_FRAMESIZE, nframes)
    ctx.upload_i32(frames, FULL_FRAMESIZE)
    g = gpu_cfg(cfg)
    t = time.time()
    ctx.analyse(g)
    recs, prof = ctx.encode_frames(g)
    dt = time.time() - t
    kt = ctx.kernel_times()
    dec, _ = ctx.decode_frames(recs[: min(4, nframes)], FULL_FRAMESIZE)      # GPU decoder as the lossless check of a few frames
    ok_dec = all(np.array_equal(d, f) for d, f in zip(dec, frames))
    ctx.close()
    nsamp = sum(f.size for f in frames)
    same = (hashlib.sha256(recs[0]).digest() == golden[f"cfg/{name}/record_sha256"].tobytes()
            and len(recs[0]) == int(golden[f"cfg/{name}/record_len"][0]) and np.array_equal(prof[0], golden[f"cfg/{name}/profile"]))
    full = f"cfg/{name}/wall_seconds_8_threads" in golden
    cpu_s = float(golden[f"cfg/{name}/wall_seconds_8_threads" if full else f"cfg/{name}/cpu_seconds"][0])
    if full:       # round 5: the presets' full evaluation counts; the reference timed with its own threading for dds,8 (8 threads) in the build container
        return {"config": name, "frames": nframes, "channels": nch, "bits": bits, "frame_seconds": 20, "maxnfunc": cfg.maxnfunc, "dds_n": cfg.num_threads,
                "fraction": cfg.fraction, "cost": int(cfg.cost), "seconds": dt, "MSamples_per_s": nsamp / dt / 1e6, "bps": 8 * sum(len(r) for r in recs) / nsamp,
                "record0_equals_reference": bool(same), "gpu_decoder_roundtrip_ok": bool(ok_dec), "kernel_ms": {k: round(v["ms"], 1) for k, v in kt.items()},
                "reference_wall_seconds_frame0_build_container_8_threads": cpu_s, "reference_MSamples_per_s_build_container_8_threads": raw0.size / cpu_s / 1e6,
                "speedup_vs_reference_8_threads_build_container": (nsamp / dt) / (raw0.size / cpu_s)}
    return {"config": name, "frames": nframes, "channels": nch, "bits": bits, "frame_seconds": 20, "maxnfunc": cfg.maxnfunc, "dds_n": cfg.num_threads,
            "fraction": cfg.fraction, "cost": int(cfg.cost), "seconds": dt, "MSamples_per_s": nsamp / dt / 1e6, "bps": 8 * sum(len(r) for r in recs) / nsamp,
            "record0_equals_reference": bool(same), "gpu_decoder_roundtrip_ok": bool(ok_dec),
            "kernel_ms": {k: round(v["ms"], 1) for k, v in kt.items()},
            "reference_cpu_seconds_frame0_build_container_1_core": cpu_s, "reference_cpu_MSamples_per_s_build_container": raw0.size / cpu_s / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames-best", type=int, default=16)
    ap.add_argument("--frames-vh", type=int, default=32)
    ap.add_argument("--full", default="", help="comma list of config34_full_cases names (the presets' full evaluation counts: vh_m8_e300, vh_s16_e300, best_s16_e1000) "
                                               "run INSTEAD of the reduced cases, against tests/golden/ref_golden_r6.npz")
    a = ap.parse_args()
    if a.full:
        from golden_cases import config34_full_cases
        g6 = np.load(os.path.join(ROOT, "tests", "golden", "ref_golden_r6.npz"))
        spec = {"vh_m8_e300": (a.frames_vh, 1, 8, 3100), "vh_s16_e300": (a.frames_vh, 2, 16, 3200), "best_s16_e1000": (a.frames_best, 2, 16, 3000), "best_s16_e100": (a.frames_best, 2, 16, 3000)}
        for name in a.full.split(","):
            nf, nch, bits, seed0 = spec[name]
            print(json.dumps(run(name, nf, nch, bits, seed0, g6, cases=config34_full_cases())), flush=True)
        return
    golden = np.load(os.path.join(ROOT, "tests", "golden", "ref_golden_r5.npz"))
    print(json.dumps(run("best_s16_e17", a.frames_best, 2, 16, 3000, golden)), flush=True)
    print(json.dumps(run("vh_m8_e25", a.frames_vh, 1, 8, 3100, golden)), flush=True)
    print(json.dumps(run("vh_s16_e25", a.frames_vh, 2, 16, 3200, golden)), flush=True)


if __name__ == "__main__":
    main()
