"""ctypes bindings for the two TEST-ONLY checkers:

  * ``Checker("orc")`` -> oracle/liboracle.so   (this repo's CPU restatement)
  * ``Checker("ref")`` -> oracle/_ref/libsacref.so (genuine reference classes; optional)

Both export the same functions with a different prefix, so every call can be run against
both and compared.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_double, c_int, c_void_p, POINTER, byref

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

COST_L1, COST_RMS, COST_ENTROPY, COST_GOLOMB, COST_BITPLANE = 0, 1, 2, 3, 4


class FrameCfg(ctypes.Structure):
    _fields_ = [("optimize", c_int), ("fraction", c_double), ("maxnfunc", c_int),
                ("num_threads", c_int), ("sigma", c_double), ("optk", c_int), ("cost", c_int),
                ("reset", c_int), ("sparse_pcm", c_int), ("zero_mean", c_int)]


def frame_cfg(mode: str = "normal", num_threads: int = 0, reset: int = 1, sparse_pcm: int = 1,
              zero_mean: int = 1, fraction=None, maxnfunc=None, cost=COST_ENTROPY, optk=4,
              sigma=None) -> FrameCfg:
    presets = {  # cmdline.cpp:127-156
        "normal": (0, 0.0, 0, 0.2, COST_ENTROPY),
        "high": (1, 0.1, 100, 0.20, COST_ENTROPY),
        "veryhigh": (1, 0.2, 300, 0.25, COST_ENTROPY),
        "extrahigh": (1, 0.2, 600, 0.25, COST_ENTROPY),
        "best": (1, 0.5, 1000, 0.25, COST_BITPLANE),
        "insane": (1, 0.5, 1500, 0.25, COST_BITPLANE),
    }
    o, f, e, s, c = presets[mode]
    if cost != COST_ENTROPY:
        c = cost
    return FrameCfg(o, f if fraction is None else fraction, e if maxnfunc is None else maxnfunc,
                    num_threads, s if sigma is None else sigma, optk, c, reset, sparse_pcm,
                    zero_mean)


def _vp(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_void_p)


def build_oracle(force: bool = False) -> None:
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "sac_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def ref_available() -> bool:
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libsacref.so"))


class Checker:
    def __init__(self, prefix: str = "orc"):
        self.prefix = prefix
        if prefix == "orc":
            build_oracle()
            path = os.path.join(ORACLE_DIR, "liboracle.so")
        elif prefix == "ref":
            path = os.path.join(ORACLE_DIR, "_ref", "libsacref.so")
        else:
            raise ValueError(prefix)
        self.lib = ctypes.CDLL(path)
        for name in ("cost", "remap", "reflect", "dds_quadratic", "dot", "s2pow"):
            getattr(self.lib, f"{prefix}_{name}").restype = c_double

    def f(self, name):
        return getattr(self.lib, f"{self.prefix}_{name}")

    # ---- profile / tables
    def profile(self) -> np.ndarray:
        out = np.zeros((58, 3), np.float32)
        self.f("profile")(_vp(out))
        return out

    def domain_tables(self):
        fwd = np.zeros(32768, np.int32)
        inv = np.zeros(4095, np.int32)
        self.f("domain_tables")(_vp(fwd), _vp(inv))
        return fwd, inv

    # ---- predictor
    def predict_frame(self, samples, stats, coefs, start, n, optimize, optk=4, framesize=None):
        samples = np.ascontiguousarray(samples, np.int32)
        nch, total = samples.shape
        stats = np.ascontiguousarray(stats, np.int32)
        coefs = np.ascontiguousarray(coefs, np.float32)
        err = np.zeros((nch, n), np.int32)
        pred = np.zeros((nch, n), np.int32)
        self.f("predict_frame")(nch, framesize or max(total, 16), total, _vp(samples), _vp(stats),
                                _vp(coefs), start, n, int(optimize), optk, _vp(err), _vp(pred))
        return err, pred

    def predict_trace(self, samples, stats, coefs, start, n, optimize, optk=4):
        samples = np.ascontiguousarray(samples, np.int32)
        nch, total = samples.shape
        stats = np.ascontiguousarray(stats, np.int32)
        coefs = np.ascontiguousarray(coefs, np.float32)
        pd = np.zeros((nch, n)); plpc = np.zeros((nch, n)); plms = np.zeros((nch, n))
        err = np.zeros((nch, n), np.int32)
        self.f("predict_trace")(nch, total, _vp(samples), _vp(stats), _vp(coefs), start, n,
                                int(optimize), optk, _vp(pd), _vp(plpc), _vp(plms), _vp(err))
        return pd, plpc, plms, err

    # ---- costs
    def cost(self, kind, buf) -> float:
        buf = np.ascontiguousarray(buf, np.int32)
        return self.f("cost")(kind, _vp(buf), buf.size)

    # ---- coder
    def bitplane_encode(self, s2u, maxbpn) -> bytes:
        s2u = np.ascontiguousarray(s2u, np.int32)
        out = np.zeros(s2u.size * 4 + 1024, np.uint8)
        n = self.f("bitplane_encode")(_vp(s2u), s2u.size, maxbpn, _vp(out), out.size)
        assert n >= 0
        return out[:n].tobytes()

    def bitplane_trace(self, s2u, maxbpn, maxdec):
        s2u = np.ascontiguousarray(s2u, np.int32)
        p1 = np.zeros(maxdec, np.uint16); bits = np.zeros(maxdec, np.uint8)
        cnt = self.f("bitplane_trace")(_vp(s2u), s2u.size, maxbpn, _vp(p1), _vp(bits), maxdec)
        m = min(cnt, maxdec)
        return cnt, p1[:m], bits[:m]

    def bitplane_decode(self, data: bytes, n, maxbpn) -> np.ndarray:
        buf = np.frombuffer(data, np.uint8).copy()
        out = np.zeros(n, np.int32)
        self.f("bitplane_decode")(_vp(buf), buf.size, n, maxbpn, _vp(out))
        return out

    def rangecoder_encode(self, p1s, bits) -> bytes:
        p1s = np.ascontiguousarray(p1s, np.uint16); bits = np.ascontiguousarray(bits, np.uint8)
        out = np.zeros(p1s.size * 2 + 64, np.uint8)
        n = self.f("rangecoder_encode")(_vp(p1s), _vp(bits), p1s.size, _vp(out), out.size)
        return out[:n].tobytes()

    # ---- remap
    def remap(self, raw, pred, error):
        raw = np.ascontiguousarray(raw, np.int32); pred = np.ascontiguousarray(pred, np.int32)
        error = np.ascontiguousarray(error, np.int32)
        n = raw.size
        s2u_map = np.zeros(n, np.int32); mb = c_int(0)
        ul = np.zeros(32769, np.uint8); uh = np.zeros(32769, np.uint8)
        r = self.f("remap")(_vp(raw), n, _vp(pred), _vp(error), _vp(s2u_map), byref(mb), _vp(ul), _vp(uh))
        return r, s2u_map, mb.value, ul, uh

    def mapencode(self, ul, uh) -> bytes:
        out = np.zeros(1 << 16, np.uint8)
        n = self.f("mapencode")(_vp(np.ascontiguousarray(ul, np.uint8)), _vp(np.ascontiguousarray(uh, np.uint8)), _vp(out), out.size)
        return out[:n].tobytes()

    def analyse(self, raw):
        raw = np.ascontiguousarray(raw, np.int32)
        out = np.zeros(3, np.int32)
        self.f("analyse")(_vp(raw), raw.size, _vp(out))
        return out  # mean, min, max

    # ---- adaptive sub-frame split (Codec::Analyse)
    def plan_subframes(self, pcm, blocksamples, min_frame_length, samples_read=None):
        """pcm [nch, n] int32 (raw) -> list of (start, length, state)."""
        pcm = np.ascontiguousarray(pcm, np.int32)
        nch, n = pcm.shape
        sr = n if samples_read is None else samples_read
        out = np.zeros(3 * 64, np.int32)
        fn = self.f("plan_subframes"); fn.restype = ctypes.c_int
        cnt = fn(nch, sr, _vp(pcm), ctypes.c_longlong(n), blocksamples, min_frame_length, _vp(out), 64)
        assert 0 <= cnt <= 64
        return [tuple(int(x) for x in out[3 * i: 3 * i + 3]) for i in range(cnt)]

    def sparse_cost(self, buf):
        buf = np.ascontiguousarray(buf, np.int32)
        out = np.zeros(2)
        self.f("sparse_cost")(_vp(buf), buf.size, _vp(out))
        return float(out[0]), float(out[1])

    # ---- .sac container, genuine reader (ref only)
    def read_sac(self, path, nch_hint=2, cap_samples=1 << 22):
        hdr = np.zeros(6, np.int32); md5 = np.zeros(16, np.uint8); meta = np.zeros(1 << 16, np.uint8)
        pcm = np.zeros(cap_samples, np.int32)
        fn = self.f("read_sac"); fn.restype = ctypes.c_int
        nf = fn(ctypes.c_char_p(path.encode()), _vp(hdr), _vp(md5), _vp(meta), meta.size, _vp(pcm), ctypes.c_longlong(pcm.size))
        assert nf > 0, nf
        nch, total = int(hdr[0]), int(hdr[3])
        return dict(numchannels=nch, samplerate=int(hdr[1]), bitspersample=int(hdr[2]), numsamples=total, max_framelen=int(hdr[4]),
                    metadatasize=int(hdr[5])), md5.tobytes(), meta[: int(hdr[5])].tobytes(), pcm[: nch * total].reshape(nch, total).copy(), nf

    # ---- search helpers
    def rng(self, kinds, args=None):
        kinds = np.ascontiguousarray(kinds, np.int32)
        args = np.zeros(kinds.size) if args is None else np.ascontiguousarray(args, np.float64)
        out = np.zeros(kinds.size)
        self.f("rng")(kinds.size, _vp(kinds), _vp(args), _vp(out))
        return out

    def gen_norm(self, x, xmin, xmax, r, n):
        out = np.zeros(n)
        self.f("gen_norm")(c_double(x), c_double(xmin), c_double(xmax), c_double(r), n, _vp(out))
        return out

    def reflect(self, x, lo, hi):
        return self.f("reflect")(c_double(x), c_double(lo), c_double(hi))

    def ssc(self, which, lambdas, sigma0):
        lambdas = np.ascontiguousarray(lambdas, np.float64)
        out = np.zeros(lambdas.size)
        self.f("ssc")(which, lambdas.size, _vp(lambdas), c_double(sigma0), _vp(out))
        return out

    def dds_quadratic(self, xmin, xmax, xstart, center, nfunc_max, num_threads, sigma):
        xmin = np.ascontiguousarray(xmin, np.float64); xmax = np.ascontiguousarray(xmax, np.float64)
        xstart = np.ascontiguousarray(xstart, np.float64); center = np.ascontiguousarray(center, np.float64)
        xb = np.zeros(xmin.size); tc = np.zeros(nfunc_max)
        best = self.f("dds_quadratic")(xmin.size, _vp(xmin), _vp(xmax), _vp(xstart), _vp(center),
                                       nfunc_max, num_threads, c_double(sigma), _vp(xb), _vp(tc))
        return best, xb, tc

    def search_quadratic(self, search, xmin, xmax, xstart, center, nfunc_max, sigma, num_threads=1):
        """DriverDDS / DriverDE / DriverCMA (search 0 / 1 / 2; oracle/ref_driver.cpp) on f(x) = sum |x_i - c_i| / (i+1);
        genuine-reference checker only.  -> (best cost, best point, cost of every evaluation)"""
        assert self.prefix == "ref"
        xmin = np.ascontiguousarray(xmin, np.float64); xmax = np.ascontiguousarray(xmax, np.float64)
        xstart = np.ascontiguousarray(xstart, np.float64); center = np.ascontiguousarray(center, np.float64)
        xb = np.zeros(xmin.size); tc = np.zeros(nfunc_max + 64); ne = ctypes.c_int(0)
        fn = self.lib.ref_search_quadratic; fn.restype = c_double
        best = fn(int(search), xmin.size, _vp(xmin), _vp(xmax), _vp(xstart), _vp(center), int(nfunc_max), int(num_threads),
                  c_double(sigma), _vp(xb), _vp(tc), byref(ne))
        return best, xb, tc[: ne.value].copy()

    # ---- whole frame
    def encode_frame(self, raw, cfg: FrameCfg, framesize, profile=None, trace=False, search=0):
        """search: FrameCoder::SearchMethod 0 DDS, 1 DE, 2 CMA (the two latter with the genuine-reference checker only)."""
        getattr(self.lib, f"{self.prefix}_set_search_method")(int(search))
        raw = np.ascontiguousarray(raw, np.int32)
        nch, n = raw.shape
        prof = self.profile()[:, 2].copy() if profile is None else np.ascontiguousarray(profile, np.float32).copy()
        out = np.zeros(n * nch * 4 + 65536 * 2 + 4096, np.uint8)
        info = np.zeros(6, np.int32)
        ntr = max(cfg.maxnfunc, 1) + 32          # ref_driver: up to maxnfunc + 32 evaluations are traced (DE start-up population)
        tc = np.full(ntr, np.nan) if trace else None
        tg = np.zeros((ntr, 58), np.float32) if trace else None
        m = self.f("encode_frame")(nch, framesize, n, _vp(raw), byref(cfg), _vp(prof), _vp(out), out.size,
                                   _vp(tc), _vp(tg), _vp(info))
        assert m > 0, m
        res = {"record": out[:m].tobytes(), "profile": prof, "info": info.reshape(2, 3)[:nch]}
        if trace:
            nev = int(np.count_nonzero(~np.isnan(tc))) if search else max(cfg.maxnfunc, 1)
            res["trace_cost"] = tc[:nev]; res["trace_coefs"] = tg[:nev]
        return res

    def decode_frame(self, rec: bytes, nch, framesize):
        buf = np.frombuffer(rec, np.uint8).copy()
        if nch == 2 and buf.size >= 4 + 58 * 4:
            # the reference's stereo channel loop (libsac.cpp:128-140, :166-198) never terminates when the frame is shorter
            # than nS1 = |round(coef 27)|; the oracle restates that loop, so refuse instead of hanging the caller
            ns = int(np.frombuffer(buf[:4].tobytes(), "<u4")[0])
            ns1 = abs(int(np.round(np.frombuffer(buf[4: 4 + 58 * 4].tobytes(), "<f4")[27])))
            if ns < ns1:
                raise ValueError(f"stereo frame of {ns} samples with nS1 = {ns1}: the reference decoder does not terminate on it")
        out = np.zeros((nch, framesize), np.int32)
        coefs = np.zeros(58, np.float32)
        # out is planar [nch][n]; decode into a flat buffer then reshape
        flat = np.zeros(nch * framesize, np.int32)
        n = self.f("decode_frame")(_vp(buf), buf.size, nch, framesize, _vp(flat), framesize, _vp(coefs))
        if n < 0:
            raise RuntimeError(f"decode_frame failed: {n}")
        return flat[: nch * n].reshape(nch, n).copy(), coefs

    # ---- math probes
    def dot(self, x, y):
        x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
        return self.f("dot")(_vp(x), _vp(y), x.size)

    def s2pow(self, x, p):
        x = np.ascontiguousarray(x, np.float64); p = np.ascontiguousarray(p, np.float64)
        return self.f("s2pow")(_vp(x), _vp(p), x.size)

    def ldlt(self, A, nu, b):
        A = np.ascontiguousarray(A, np.float64); b = np.ascontiguousarray(b, np.float64)
        w = np.zeros(b.size)
        ok = self.f("ldlt")(_vp(A), b.size, c_double(nu), _vp(b), _vp(w))
        return ok, w


def center_frame(raw: np.ndarray, zero_mean: bool = True):
    """FrameCoder::Predict's per-channel stats + mean removal (libsac.cpp:445-459) in numpy:
    returns (mean-removed samples, stats[nch,3]={min,max,mean})."""
    raw = np.asarray(raw, np.int64)
    nch, n = raw.shape
    stats = np.zeros((nch, 3), np.int32)
    out = np.zeros((nch, n), np.int32)
    for ch in range(nch):
        mean = int(np.floor(raw[ch].sum() / float(n))) if zero_mean else 0
        out[ch] = raw[ch] - mean
        stats[ch] = (raw[ch].min() - mean, raw[ch].max() - mean, mean)
    return out, stats
