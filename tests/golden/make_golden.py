"""Generate tests/golden/ref_golden.npz from the GENUINE reference (oracle/_ref/libsacref.so,
built by `make -C oracle ref` from /root/reference/src).  Run in the build container only:

    python tests/golden/make_golden.py

The file holds inputs and expected outputs only (no reference source text).
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from golden_cases import FRAMESIZE, frame_cases, trace_cases  # noqa: E402
from oracle_api import Checker, center_frame  # noqa: E402


def main_subframes():
    """tests/golden/subframes_golden.npz: Codec::Analyse sub-frame lists and SparsePCM costs."""
    from golden_cases import subframe_cases
    R = Checker("ref")
    out = {}
    for name, (pcm, blk, min_len) in subframe_cases().items():
        out[f"{name}/pcm"] = pcm.astype(np.int16)
        out[f"{name}/args"] = np.array([blk, min_len], np.int32)
        out[f"{name}/subframes"] = np.array(R.plan_subframes(pcm, blk, min_len), np.int32).reshape(-1, 3)
        out[f"{name}/cost_block0"] = np.array([R.sparse_cost(pcm[ch, :blk]) for ch in range(pcm.shape[0])])
    path = os.path.join(HERE, "subframes_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main_r2():
    """tests/golden/ref_golden_r2.npz: canonical-order cascade traces for every layout class and the full-size
    (882 000-sample) frame records of the headline configuration (record bytes as SHA-256 + length + profile)."""
    import hashlib
    from golden_cases import FULL_FRAMESIZE, fullsize_cases, trace_cases_r2
    R = Checker("ref")
    out = {}
    P = R.profile()
    for name, (raw, coefs, opt, start, n) in trace_cases_r2(P).items():
        smp, stats = center_frame(raw)
        pd, plpc, plms, err = R.predict_trace(smp, stats, coefs, start, n, opt)
        out[f"trace/{name}/raw"] = raw.astype(np.int16)
        out[f"trace/{name}/coefs"] = coefs
        out[f"trace/{name}/plpc"] = plpc
        out[f"trace/{name}/plms"] = plms
        out[f"trace/{name}/err"] = err
    for name, (raw, cfg) in fullsize_cases().items():
        r = R.encode_frame(raw, cfg, FULL_FRAMESIZE)
        out[f"full/{name}/raw_sha256"] = np.frombuffer(hashlib.sha256(raw.astype(np.int16).tobytes()).digest(), np.uint8)
        out[f"full/{name}/record_sha256"] = np.frombuffer(hashlib.sha256(r["record"]).digest(), np.uint8)
        out[f"full/{name}/record_len"] = np.array([len(r["record"])], np.int64)
        out[f"full/{name}/profile"] = r["profile"]
        print(name, len(r["record"]), "bytes")
    # the headline configuration itself (100 evaluations) on frame 0 of bench.py's batch: search trace + record
    from oracle_api import frame_cfg
    name, (raw, _) = next(iter(fullsize_cases().items()))
    r = R.encode_frame(raw, frame_cfg("high", num_threads=8), FULL_FRAMESIZE, trace=True)
    out["full100/trace_cost"] = r["trace_cost"]
    out["full100/trace_coefs"] = r["trace_coefs"]
    out["full100/profile"] = r["profile"]
    out["full100/record_sha256"] = np.frombuffer(hashlib.sha256(r["record"]).digest(), np.uint8)
    out["full100/record_len"] = np.array([len(r["record"])], np.int64)
    print("full100", len(r["record"]), "bytes")
    path = os.path.join(HERE, "ref_golden_r2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main_r3():
    """tests/golden/ref_golden_r3.npz (round 3): warm-start chains (reset=0) of consecutive frames."""
    from golden_cases import chain_cases
    R = Checker("ref")
    out = {}
    for name, (frames, cfg) in chain_cases().items():
        prof = None
        for f, raw in enumerate(frames):
            r = R.encode_frame(raw, cfg, FRAMESIZE, profile=prof)
            prof = r["profile"]
            out[f"chain/{name}/{f}/raw"] = raw.astype(np.int16)
            out[f"chain/{name}/{f}/record"] = np.frombuffer(r["record"], np.uint8)
            out[f"chain/{name}/{f}/profile"] = prof
    path = os.path.join(HERE, "ref_golden_r3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main_r4():
    """tests/golden/ref_golden_r4.npz (round 3): the DE and CMA searchers (--opt-cfg=de / cma) -- traces on an analytic
    function and frame records, from oracle/_ref (DriverDE / DriverCMA around the genuine Opt, Cholesky, SSC1)."""
    from golden_cases import search_cases, search_quadratic_cases, search_quadratic_inputs
    R = Checker("ref")
    out = {}
    for name, (search, ndim, nmax, sigma, seed) in search_quadratic_cases().items():
        lo, hi, xs, cen = search_quadratic_inputs(ndim, seed)
        best, xb, tc = R.search_quadratic(search, lo, hi, xs, cen, nmax, sigma)
        out[f"quad/{name}/xbest"] = xb; out[f"quad/{name}/trace"] = tc
    for name, (raw, cfg, search) in search_cases().items():
        r = R.encode_frame(raw, cfg, FRAMESIZE, trace=True, search=search)
        out[f"search/{name}/raw"] = raw.astype(np.int16)
        out[f"search/{name}/record"] = np.frombuffer(r["record"], np.uint8)
        out[f"search/{name}/profile"] = r["profile"]
        out[f"search/{name}/trace_cost"] = r["trace_cost"]
        out[f"search/{name}/trace_coefs"] = r["trace_coefs"]
        print(name, len(r["record"]), "bytes")
    # 24-bit material (SURVEY 8f rank 3 remainder): raw as int32
    from golden_cases import wide_cases
    for name, (raw, cfg) in wide_cases().items():
        r = R.encode_frame(raw, cfg, FRAMESIZE, trace=True)
        out[f"wide/{name}/raw"] = raw.astype(np.int32)
        out[f"wide/{name}/record"] = np.frombuffer(r["record"], np.uint8)
        out[f"wide/{name}/profile"] = r["profile"]
        out[f"wide/{name}/trace_cost"] = r["trace_cost"]
        out[f"wide/{name}/trace_coefs"] = r["trace_coefs"]
        print(name, len(r["record"]), "bytes", r["info"].tolist())
    path = os.path.join(HERE, "ref_golden_r4.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main_r5():
    """tests/golden/ref_golden_r5.npz (round 3): BASELINE configs[3] (--best) and configs[4] (--veryhigh, 8-bit mono and
    16-bit stereo) on full-size frames with a reduced evaluation count: record SHA-256 + length + chosen profile + search costs."""
    import hashlib, time
    from golden_cases import FULL_FRAMESIZE, config34_cases
    R = Checker("ref")
    out = {}
    for name, (raw, cfg) in config34_cases().items():
        t = time.time()
        r = R.encode_frame(raw, cfg, FULL_FRAMESIZE, trace=True)
        out[f"cfg/{name}/raw_sha256"] = np.frombuffer(hashlib.sha256(raw.astype(np.int16).tobytes()).digest(), np.uint8)
        out[f"cfg/{name}/record_sha256"] = np.frombuffer(hashlib.sha256(r["record"]).digest(), np.uint8)
        out[f"cfg/{name}/record_len"] = np.array([len(r["record"])], np.int64)
        out[f"cfg/{name}/profile"] = r["profile"]
        out[f"cfg/{name}/trace_cost"] = r["trace_cost"]
        out[f"cfg/{name}/cpu_seconds"] = np.array([time.time() - t])
        print(name, len(r["record"]), "bytes", round(time.time() - t, 1), "s on one core")
    path = os.path.join(HERE, "ref_golden_r5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main_r6(only=None):
    """tests/golden/ref_golden_r6.npz (round 5): BASELINE configs[3] / [4] with the presets' FULL evaluation counts (--veryhigh
    E = 300, --best E = 1000), one 882 000-sample frame each, genuine reference objects with the reference's own threading for
    --opt-cfg=dds,8 (Opt::eval_points_mt on 8 threads: same results as serial evaluation, an eighth of the wall time)."""
    import hashlib, time
    from golden_cases import FULL_FRAMESIZE, config34_full_cases
    R = Checker("ref")
    R.lib.ref_set_parallel_eval(1)
    path = os.path.join(HERE, "ref_golden_r6.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name, (raw, cfg) in config34_full_cases().items():
        if only and name not in only:
            continue
        t = time.time()
        r = R.encode_frame(raw, cfg, FULL_FRAMESIZE, trace=True)
        out[f"cfg/{name}/raw_sha256"] = np.frombuffer(hashlib.sha256(raw.astype(np.int16).tobytes()).digest(), np.uint8)
        out[f"cfg/{name}/record_sha256"] = np.frombuffer(hashlib.sha256(r["record"]).digest(), np.uint8)
        out[f"cfg/{name}/record_len"] = np.array([len(r["record"])], np.int64)
        out[f"cfg/{name}/profile"] = r["profile"]
        out[f"cfg/{name}/trace_cost"] = r["trace_cost"]
        out[f"cfg/{name}/wall_seconds_8_threads"] = np.array([time.time() - t])
        print(name, len(r["record"]), "bytes", round(time.time() - t, 1), "s with 8 evaluation threads", flush=True)
        np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main_r7(only=None):
    """tests/golden/ref_golden_r7.npz (round 6): searches at the presets' full lengths (golden_cases.preset_length_cases): record
    bytes (small frame) or SHA-256 (full-size frame), chosen profile, every search cost."""
    import hashlib, time
    from golden_cases import preset_length_cases
    R = Checker("ref")
    path = os.path.join(HERE, "ref_golden_r7.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name, (raw, cfg, fs) in preset_length_cases().items():
        if only and name not in only:
            continue
        R.lib.ref_set_parallel_eval(1 if cfg.num_threads > 1 else 0)
        t = time.time()
        r = R.encode_frame(raw, cfg, fs, trace=True)
        out[f"cfg/{name}/raw_sha256"] = np.frombuffer(hashlib.sha256(raw.astype(np.int16).tobytes()).digest(), np.uint8)
        out[f"cfg/{name}/record_sha256"] = np.frombuffer(hashlib.sha256(r["record"]).digest(), np.uint8)
        out[f"cfg/{name}/record_len"] = np.array([len(r["record"])], np.int64)
        if raw.size <= 100000:
            out[f"cfg/{name}/record"] = np.frombuffer(r["record"], np.uint8)
        out[f"cfg/{name}/profile"] = r["profile"]
        out[f"cfg/{name}/trace_cost"] = r["trace_cost"]
        out[f"cfg/{name}/wall_seconds"] = np.array([time.time() - t])
        print(name, len(r["record"]), "bytes", len(r["trace_cost"]), "costs", round(time.time() - t, 1), "s", flush=True)
        np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main():
    R = Checker("ref")
    out = {}
    P = R.profile()
    out["profile"] = P
    fwd, inv = R.domain_tables()
    out["domain_crc"] = np.array([zlib.crc32(fwd.tobytes()), zlib.crc32(inv.tobytes())], np.uint64)
    out["domain_fwd_sample"] = fwd[::257].copy()
    out["domain_inv_sample"] = inv[::31].copy()

    # whole-frame records (+ search traces)
    for name, (raw, cfg) in frame_cases().items():
        r = R.encode_frame(raw, cfg, FRAMESIZE, trace=True)
        out[f"frame/{name}/raw"] = raw
        out[f"frame/{name}/record"] = np.frombuffer(r["record"], np.uint8)
        out[f"frame/{name}/profile"] = r["profile"]
        out[f"frame/{name}/info"] = r["info"]
        if cfg.optimize:
            out[f"frame/{name}/trace_cost"] = r["trace_cost"]
            out[f"frame/{name}/trace_coefs"] = r["trace_coefs"]

    # predictor traces (fp64 bit patterns)
    for name, (raw, coefs, opt, start, n) in trace_cases(P).items():
        smp, stats = center_frame(raw)
        pd, plpc, plms, err = R.predict_trace(smp, stats, coefs, start, n, opt)
        out[f"trace/{name}/raw"] = raw
        out[f"trace/{name}/coefs"] = coefs
        out[f"trace/{name}/pd"] = pd
        out[f"trace/{name}/plpc"] = plpc
        out[f"trace/{name}/plms"] = plms
        out[f"trace/{name}/err"] = err

    # coder trace: first 20000 (p1,bit) decisions + bytes for one residual vector
    rng = np.random.default_rng(11)
    e = np.rint(rng.laplace(size=4000) * 150).astype(np.int32)
    u = np.where(e < 0, -2 * e, np.where(e > 0, 2 * e - 1, 0)).astype(np.int32)
    mb = int(u.max()).bit_length() - 1
    cnt, p1, bits = R.bitplane_trace(u, mb, 20000)
    out["coder/s2u"] = u
    out["coder/maxbpn"] = np.array([mb, cnt], np.int64)
    out["coder/p1"] = p1
    out["coder/bits"] = bits
    out["coder/bytes"] = np.frombuffer(R.bitplane_encode(u, mb), np.uint8)

    # costs
    e2 = np.rint(rng.laplace(size=3000) * 40).astype(np.int32)
    out["cost/err"] = e2
    out["cost/values"] = np.array([R.cost(k, e2) for k in range(5)])

    # RNG / search helpers (pins libstdc++ <random> behaviour on whatever host runs the tests)
    kinds = rng.integers(0, 3, 1000).astype(np.int32)
    out["rng/kinds"] = kinds
    out["rng/values"] = R.rng(kinds, np.full(1000, 55.0))
    out["rng/gen_norm"] = R.gen_norm(0.3, 0.0, 1.0, 0.2, 300)
    nd = 12
    lo = np.zeros(nd); hi = np.arange(1, nd + 1) * 1.0
    for nt in (0, 4):
        best, xb, tc = R.dds_quadratic(lo, hi, hi * 0.5, hi * 0.25, 120, nt, 0.2)
        out[f"dds/q{nt}/xbest"] = xb
        out[f"dds/q{nt}/trace"] = tc

    # remap
    raw = frame_cases()["sparse16_normal"][0][0]
    smp, stats = center_frame(raw[None, :])
    err, pred = R.predict_frame(smp, stats, P[:, 2].copy(), 0, raw.size, 0)
    r, s2u_map, mbm, ul, uh = R.remap(raw, pred[0], err[0])
    out["remap/ratio"] = np.array([r])
    out["remap/s2u_map"] = s2u_map
    out["remap/maxbpn"] = np.array([mbm])
    out["remap/mapbytes"] = np.frombuffer(R.mapencode(ul, uh), np.uint8)

    path = os.path.join(HERE, "ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    if "--subframes" in sys.argv:
        main_subframes()
    elif "--r2" in sys.argv:
        main_r2()
    elif "--r3" in sys.argv:
        main_r3()
    elif "--r4" in sys.argv:
        main_r4()
    elif "--r5" in sys.argv:
        main_r5()
    elif "--r7" in sys.argv:
        main_r7([a for a in sys.argv[1:] if not a.startswith("--")] or None)
    elif "--r6" in sys.argv:
        main_r6([a for a in sys.argv[1:] if not a.startswith("--")] or None)
    else:
        main()
