"""Shared definition of the parity cases (inputs are regenerated from seeds; the golden file
also stores the raw PCM so the fixtures do not depend on numpy's RNG staying stable)."""
from __future__ import annotations

import numpy as np

from oracle_api import COST_BITPLANE, frame_cfg
from sac_amd.synth import synth_pcm

RATE = 8000
FRAMESIZE = 20 * RATE  # max_framelen(20 s) * rate, libsac.cpp:784


def frame_cases():
    """name -> (raw PCM [nch,n] int32, FrameCfg)."""
    c = {}
    c["s16_normal"] = (synth_pcm(6000, 2, 1, RATE), frame_cfg("normal"))
    c["m16_normal"] = (synth_pcm(6000, 1, 2, RATE), frame_cfg("normal"))
    c["m8_normal"] = (synth_pcm(6000, 1, 3, RATE, bits=8), frame_cfg("normal"))
    c["sparse16_normal"] = (synth_pcm(6000, 1, 4, RATE, sparse_bits=10), frame_cfg("normal"))
    c["sparse16s_normal"] = (synth_pcm(6000, 2, 5, RATE, sparse_bits=9), frame_cfg("normal"))
    c["s16_high_single"] = (synth_pcm(8000, 2, 6, RATE), frame_cfg("high", maxnfunc=24))
    c["s16_high_mt4"] = (synth_pcm(8000, 2, 7, RATE), frame_cfg("high", num_threads=4, maxnfunc=24))
    c["m16_bpncost_mt4"] = (synth_pcm(4000, 1, 8, RATE),
                            frame_cfg("high", num_threads=4, maxnfunc=10, cost=COST_BITPLANE, fraction=0.02))
    c["s16_raw_nosparse"] = (synth_pcm(5000, 2, 9, RATE) + 300, frame_cfg("normal", sparse_pcm=0, zero_mean=0))
    c["tiny_stereo"] = (synth_pcm(40, 2, 10, RATE), frame_cfg("normal"))
    c["silence_mono"] = (np.zeros((1, 500), np.int32), frame_cfg("normal"))
    return c


def rand_profile(P, rng, cap=True, scale=1.0):
    """Random point of the search box around the default profile (P = [58,3] vmin,vmax,vdef)."""
    g = P[:, 2].astype(np.float64).copy()
    for i in range(58):
        lo, hi = float(P[i, 0]), float(P[i, 1])
        if rng.random() < 0.7:
            g[i] = np.clip(g[i] + rng.standard_normal() * 0.2 * scale * (hi - lo), lo, hi)
    if cap:  # keep the CPU checkers fast
        for i, c in zip([28, 29, 30, 37, 31, 32, 33, 38], [600, 200, 64, 16, 600, 200, 64, 16]):
            g[i] = min(g[i], c)
    return g.astype(np.float32)


def trace_cases(P):
    """name -> (raw [nch,n], coefs[58], optimize flag, start, n_window)."""
    rng = np.random.default_rng(5)
    c = {}
    c["tr_s16_default_k1"] = (synth_pcm(1500, 2, 20, RATE), P[:, 2].copy(), 0, 0, 1500)
    c["tr_s16_default_k4_window"] = (synth_pcm(2500, 2, 21, RATE), P[:, 2].copy(), 1, 700, 1200)
    c["tr_m16_rand_k1"] = (synth_pcm(1500, 1, 22, RATE), rand_profile(P, rng), 0, 0, 1500)
    g = rand_profile(P, rng); g[27] = -11.0; g[9] = 7.0  # ch_ref swap + nM0>0
    c["tr_s16_swap_k4"] = (synth_pcm(1500, 2, 23, RATE), g, 1, 0, 1500)
    g = rand_profile(P, rng); g[27] = 0.0; g[26] = 0.0
    c["tr_s16_nS1_0"] = (synth_pcm(1200, 2, 24, RATE), g, 0, 0, 1200)
    g = rand_profile(P, rng); g[24] = 32; g[9] = 32; g[25] = 32; g[26] = 32; g[27] = 32; g[41] = 10
    c["tr_s16_maxols"] = (synth_pcm(900, 2, 25, RATE), g, 1, 0, 900)
    c["tr_m8_default"] = (synth_pcm(1500, 1, 26, RATE, bits=8), P[:, 2].copy(), 0, 0, 1500)
    return c


def trace_cases_r2(P):
    """Round-2 predictor traces (final pass, k = 1): stage lengths that exercise every canonical-order cascade
    layout (512-lane classes), the transform_reduce tails (n % 8, n % 4 != 0) and the n < 8 stages."""
    c = {}

    def prof(t0, t1=None):
        g = P[:, 2].copy()
        g[28], g[29], g[30], g[37] = t0
        g[31], g[32], g[33], g[38] = t1 if t1 is not None else t0
        return g
    c["tr_m16_taps_k1_512"] = (synth_pcm(900, 1, 40, RATE), prof((3001, 1500, 901, 300)), 0, 0, 900)
    c["tr_m16_taps_k1_max"] = (synth_pcm(600, 1, 41, RATE), prof((6007, 3003, 1703, 601)), 0, 0, 600)
    c["tr_s16_taps_k1_tiny"] = (synth_pcm(1200, 2, 42, RATE), prof((256, 32, 5, 2), (263, 39, 7, 3)), 0, 0, 1200)
    c["tr_s16_taps_k1_tails"] = (synth_pcm(1000, 2, 43, RATE), prof((1283, 257, 33, 6), (2301, 1279, 767, 255)), 0, 0, 1000)
    return c


def chain_cases():
    """name -> (list of raw frames [nch,n], FrameCfg): consecutive frames of ONE file encoded with reset=0, the
    reference's default: the best profile of frame f is the search start of frame f+1 AND the profile written for
    frame f (libsac.cpp:461-466, :571)."""
    c = {}
    c["chain_s16_mt4"] = ([synth_pcm(4000, 2, 60 + f, RATE) for f in range(3)], frame_cfg("high", num_threads=4, maxnfunc=12, reset=0))
    c["chain_m16_single"] = ([synth_pcm(3000, 1, 70 + f, RATE) for f in range(2)], frame_cfg("high", num_threads=0, maxnfunc=10, reset=0))
    return c


def search_cases():
    """name -> (raw PCM, FrameCfg, search method): --opt-cfg=de / cma (FrameCoder::SearchMethod 1 / 2) on small frames.
    DE evaluates 1 + 29 start-up points, then generations of min(30, E - evaluated) trial vectors: E = 75 ends on a
    15-vector generation, E = 31 on a single one; E = 20 < 30 still evaluates the whole start-up population."""
    c = {}
    c["s16_de_e75"] = (synth_pcm(4000, 2, 81, RATE), frame_cfg("high", maxnfunc=75), 1)
    c["m16_de_e31"] = (synth_pcm(3000, 1, 82, RATE), frame_cfg("high", maxnfunc=31, sigma=0.15), 1)
    c["m16_de_e20"] = (synth_pcm(2500, 1, 83, RATE), frame_cfg("high", maxnfunc=20), 1)
    c["s16_cma_e40"] = (synth_pcm(4000, 2, 84, RATE), frame_cfg("high", maxnfunc=40), 2)
    c["m16_cma_e25"] = (synth_pcm(3000, 1, 85, RATE), frame_cfg("high", maxnfunc=25, sigma=0.25), 2)
    return c


def wide_cases():
    """name -> (raw PCM, FrameCfg): 24-bit material (|sample| up to 2^23), encoded with --sparse-pcm=0 -- the reference's Remap
    holds 2 x 32 769 used-value flags and prints "val too large" for every wider sample, so sparse-PCM mapping is a
    16-bit feature; everything else on the path (predictor, Entropy / Bitplane cost over wide residual ranges, coder planes
    above 17, PredictLaplace beyond the 2^17 table) is exercised here."""
    c = {}
    c["s24_normal"] = (synth_pcm(5000, 2, 91, RATE, bits=24), frame_cfg("normal", sparse_pcm=0))
    c["s24_high_mt4"] = (synth_pcm(5000, 2, 91, RATE, bits=24), frame_cfg("high", num_threads=4, maxnfunc=12, sparse_pcm=0))
    c["m24_bpncost_mt4"] = (synth_pcm(3000, 1, 92, RATE, bits=24), frame_cfg("high", num_threads=4, maxnfunc=9, cost=COST_BITPLANE, fraction=0.1, sparse_pcm=0))
    loud = np.rint(np.random.default_rng(93).laplace(size=(1, 2500)) * 9e5).astype(np.int64)
    c["m24_loud_noise_high"] = (np.clip(loud, -(1 << 23), (1 << 23) - 1).astype(np.int32), frame_cfg("high", num_threads=4, maxnfunc=9, sparse_pcm=0))
    return c


def search_quadratic_cases():
    """name -> (search, ndim, nfunc_max, sigma, seed) for the searchers on the analytic test function"""
    return {"de_56_100": (1, 56, 100, 0.2, 1), "de_13_500": (1, 13, 500, 0.2, 4), "de_3_31": (1, 3, 31, 0.2, 6), "de_56_20": (1, 56, 20, 0.2, 7),
            "cma_56_100": (2, 56, 100, 0.2, 1), "cma_13_500": (2, 13, 500, 0.2, 4), "cma_7_40": (2, 7, 40, 0.15, 3)}


def search_quadratic_inputs(ndim, seed):
    rng = np.random.default_rng(seed)
    xmin = -rng.uniform(0.5, 3, ndim); xmax = rng.uniform(0.5, 100, ndim)
    xs = xmin + (xmax - xmin) * rng.uniform(0.2, 0.8, ndim); cen = xmin + (xmax - xmin) * rng.uniform(0, 1, ndim)
    return xmin, xmax, xs, cen


FULL_RATE = 44100
FULL_FRAMESIZE = 20 * FULL_RATE


def fullsize_cases():
    """name -> (raw PCM [2, 882000], FrameCfg): BASELINE configs[2] at the size the metric is quoted on (20 s stereo
    44.1 kHz, --high --opt-cfg=dds,8 --opt-reset, search window 88 200 samples) with a reduced evaluation count so
    that the CPU reference finishes in seconds per frame."""
    c = {}
    for i in range(2):
        c[f"full_s16_high_dds8_{i}"] = (synth_pcm(20 * FULL_RATE, 2, 1000 + i, FULL_RATE), frame_cfg("high", num_threads=8, maxnfunc=9))
    return c


def config34_cases():
    """BASELINE configs[3] and [4] at their real sizes with the evaluation count cut (the window and cost are what make
    them different from configs[2]): name -> (raw PCM, FrameCfg).
    best_s16: --best (CostBitplane objective over a 441 000-sample window, sigma 0.25), 17 evaluations (dds,8).
    vh_m8 / vh_s16: --veryhigh (176 400-sample window), 8-bit mono and 16-bit stereo material, 25 evaluations."""
    c = {}
    c["best_s16_e17"] = (synth_pcm(20 * FULL_RATE, 2, 3000, FULL_RATE), frame_cfg("best", num_threads=8, maxnfunc=17))
    c["vh_m8_e25"] = (synth_pcm(20 * FULL_RATE, 1, 3100, FULL_RATE, bits=8), frame_cfg("veryhigh", num_threads=8, maxnfunc=25))
    c["vh_s16_e25"] = (synth_pcm(20 * FULL_RATE, 2, 3200, FULL_RATE), frame_cfg("veryhigh", num_threads=8, maxnfunc=25))
    return c


def config34_full_cases():
    """BASELINE configs[3] and [4] with the presets' FULL evaluation counts (cmdline.cpp:127-156: --best E = 1000, --veryhigh
    E = 300), one full-size frame each (round 5): name -> (raw PCM, FrameCfg)."""
    c = {}
    c["vh_m8_e300"] = (synth_pcm(20 * FULL_RATE, 1, 3100, FULL_RATE, bits=8), frame_cfg("veryhigh", num_threads=8))
    c["vh_s16_e300"] = (synth_pcm(20 * FULL_RATE, 2, 3200, FULL_RATE), frame_cfg("veryhigh", num_threads=8))
    c["best_s16_e1000"] = (synth_pcm(20 * FULL_RATE, 2, 3000, FULL_RATE), frame_cfg("best", num_threads=8))
    c["best_s16_e100"] = (synth_pcm(20 * FULL_RATE, 2, 3000, FULL_RATE), frame_cfg("best", num_threads=8, maxnfunc=100))    # a tenth of the preset: what a GPU call can time
    return c


def preset_length_cases():
    """Round 6: searches at the presets' FULL lengths that one GPU test can afford.
    best_s16_8k_e1000: --best (cmdline.cpp:127-156: fraction 0.5, E = 1000, sigma 0.25, CostBitplane) with --opt-cfg=dds,8 on a
      reduced frame (8 kHz, 5 000 samples: the whole frame is the search window) -- the 125-generation Bitplane-cost trajectory
      at the preset's length;
    full_s16_high_single_e100: --high as the reference runs it WITHOUT --opt-cfg (OptDDS::run_single, opt/dds.cpp:33-60,
      num_threads = 0): 100 sequential evaluations on one full-size 20-s stereo frame.
    name -> (raw PCM, FrameCfg, max frame size)."""
    c = {}
    c["best_s16_8k_e1000"] = (synth_pcm(5000, 2, 3300, RATE), frame_cfg("best", num_threads=8), FRAMESIZE)
    c["full_s16_high_single_e100"] = (synth_pcm(20 * FULL_RATE, 2, 3400, FULL_RATE), frame_cfg("high", num_threads=0), FULL_FRAMESIZE)
    return c


def subframe_cases():
    """name -> (pcm [nch,n] int32 raw, blocksamples, min_frame_length): material whose 3-"second"
    blocks alternate between dense and sparse (quantised) PCM, for Codec::Analyse / PushState."""
    rate = 1000
    blk = 3 * rate
    rng = np.random.default_rng(17)
    c = {}
    specs = [("dense_only", [0, 0, 0], 1, 500, blk), ("sparse_tail", [0, 0, 0, 0, 16, 16], 2, 1945, blk),
             ("alternating", [16, 0, 4, 4, 0, 0, 0], 2, 20, blk), ("short_last_block", [0, 16, 0, 0], 1, 7, blk),
             ("min_two_blocks", [0, 4, 0, 0, 16, 0, 4, 4], 2, 1200, 2 * blk), ("all_sparse", [8, 8], 1, 0, blk)]
    for name, pattern, nch, tail, min_len in specs:
        n = len(pattern) * blk + tail
        x = synth_pcm(n, nch, int(rng.integers(1, 1 << 20)), rate) >> 5     # ~1000 distinct values: 3000-sample blocks are dense
        for i, q in enumerate(pattern + [pattern[-1]]):
            a, b = i * blk, min((i + 1) * blk, n)
            if q and a < b:
                x[:, a:b] = (x[:, a:b] // q) * q
        c[name] = (x.astype(np.int32), blk, min_len)
    return c
