"""Randomised cross-check of the CPU restatement against the genuine reference classes
(oracle/_ref/libsacref.so).  Skipped when the reference build is not present."""
import numpy as np
import pytest

from golden_cases import FRAMESIZE, RATE, rand_profile
from oracle_api import center_frame, frame_cfg
from sac_amd.synth import synth_pcm

pytestmark = pytest.mark.ref


def test_math_probes(orc, ref):
    rng = np.random.default_rng(0)
    for n in list(range(1, 20)) + [31, 32, 33, 40, 63, 64, 65]:
        for _ in range(20):
            x = rng.standard_normal(n) * rng.uniform(1, 1000)
            y = rng.standard_normal(n)
            assert orc.dot(x, y) == ref.dot(x, y)
            assert orc.s2pow(x, np.abs(y)) == ref.s2pow(x, np.abs(y))
    for n in [1, 2, 3, 5, 8, 13, 16, 32, 47, 64, 96]:
        X = rng.standard_normal((n * 3, n))
        A = X.T @ X
        b = rng.standard_normal(n)
        ok1, w1 = orc.ldlt(A, 0.05, b)
        ok2, w2 = ref.ldlt(A, 0.05, b)
        assert ok1 == ok2 and np.array_equal(w1, w2)
    # factor failure keeps the old weights
    A = -np.eye(4)
    assert orc.ldlt(A, 0.0, np.ones(4))[0] == ref.ldlt(A, 0.0, np.ones(4))[0] == 0


@pytest.mark.parametrize("seed", range(6))
def test_random_profile_traces(orc, ref, seed):
    rng = np.random.default_rng(100 + seed)
    P = ref.profile()
    nch = 1 + (seed % 2)
    raw = synth_pcm(1200, nch, 50 + seed, RATE, bits=16 if seed % 3 else 8)
    smp, stats = center_frame(raw)
    g = rand_profile(P, rng)
    opt = seed % 2
    a = orc.predict_trace(smp, stats, g, 100, 1000, opt)
    b = ref.predict_trace(smp, stats, g, 100, 1000, opt)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_uncapped_profile_short(orc, ref):
    """Large tap counts (no cap) on a short window."""
    rng = np.random.default_rng(7)
    P = ref.profile()
    g = rand_profile(P, rng, cap=False, scale=2.0)
    raw = synth_pcm(400, 2, 77, RATE)
    smp, stats = center_frame(raw)
    a = orc.predict_frame(smp, stats, g, 0, 400, 1)
    b = ref.predict_frame(smp, stats, g, 0, 400, 1)
    assert np.array_equal(a[0], b[0])


def test_frames_random(orc, ref):
    for seed, mode, nt in [(1, "normal", 0), (2, "high", 3)]:
        raw = synth_pcm(3000, 2, 200 + seed, RATE)
        cfg = frame_cfg(mode, num_threads=nt, maxnfunc=8 if mode == "high" else None)
        a = orc.encode_frame(raw, cfg, FRAMESIZE)
        b = ref.encode_frame(raw, cfg, FRAMESIZE)
        assert a["record"] == b["record"]
        d, _ = ref.decode_frame(a["record"], 2, FRAMESIZE)
        assert np.array_equal(d, raw)


def test_warm_start_profile_chain(orc, ref):
    """reset=0: the best profile of frame f seeds frame f+1 (libsac.cpp:463-466)."""
    cfg = frame_cfg("high", num_threads=2, maxnfunc=6, reset=0)
    pa = pb = None
    for f in range(2):
        raw = synth_pcm(2500, 1, 300 + f, RATE)
        a = orc.encode_frame(raw, cfg, FRAMESIZE, profile=pa)
        b = ref.encode_frame(raw, cfg, FRAMESIZE, profile=pb)
        assert a["record"] == b["record"]
        pa, pb = a["profile"], b["profile"]
        assert np.array_equal(pa, pb)


def test_subframe_plan_vs_ref(orc, ref):
    """Codec::Analyse on random dense/sparse block patterns, incl. min_frame_length > block."""
    from sac_amd.synth import synth_pcm
    rate = 2000; blk = 3 * rate
    rng = np.random.default_rng(3)
    for t in range(10):
        pat = [int(rng.choice([0, 0, 4, 16])) for _ in range(int(rng.integers(1, 8)))]
        n = len(pat) * blk + int(rng.integers(0, blk))
        x = synth_pcm(n, int(rng.integers(1, 3)), 100 + t, rate) >> 4
        for i, q in enumerate(pat + [pat[-1]]):
            a, b = i * blk, min((i + 1) * blk, n)
            if q and a < b:
                x[:, a:b] = (x[:, a:b] // q) * q
        for min_len in (blk, 2 * blk):
            assert orc.plan_subframes(x, blk, min_len) == ref.plan_subframes(x, blk, min_len)
        assert orc.sparse_cost(x[0, :blk]) == ref.sparse_cost(x[0, :blk])
