"""Synthetic PCM for tests and bench (no datasets on the box).

Spec follows SURVEY.md §8(d): per file 6 FM-modulated sinusoids (f in [80,4000] Hz,
a in [0.02,0.2], 0.1 % vibrato at 0.3 Hz) shared across channels with per-channel gain
(1.0, 0.8) + per-channel AR(2) noise (a1=1.6, a2=-0.7, sigma=0.02, gain 0.3) + white noise
sigma=0.002, normalised to 0.7 full scale and rounded to the integer grid.
"""
from __future__ import annotations

import numpy as np


def _ar2(e: np.ndarray, a1: float, a2: float) -> np.ndarray:
    try:
        from scipy.signal import lfilter

        return lfilter([1.0], [1.0, -a1, -a2], e)
    except Exception:  # pragma: no cover - scipy is in the image
        y = np.zeros_like(e)
        for i in range(len(e)):
            y[i] = e[i] + (a1 * y[i - 1] if i > 0 else 0.0) + (a2 * y[i - 2] if i > 1 else 0.0)
        return y


def synth_pcm(n: int, nch: int = 2, seed: int = 1, rate: int = 44100, bits: int = 16,
              sparse_bits: int = 0) -> np.ndarray:
    """Return int32 planar PCM [nch, n] in the signed range of `bits`.

    sparse_bits>0 quantises to that many significant bits (left-aligned) so that only a
    subset of the integer grid is used (triggers the reference's sparse-PCM mapping).
    """
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / rate
    tone = np.zeros(n)
    for _ in range(6):
        f = rng.uniform(80.0, 4000.0)
        a = rng.uniform(0.02, 0.2)
        ph = rng.uniform(0, 2 * np.pi)
        vib = 0.001 * f * np.sin(2 * np.pi * 0.3 * t + rng.uniform(0, 2 * np.pi)) / 0.3
        tone += a * np.sin(2 * np.pi * f * t + vib + ph)
    gains = [1.0, 0.8]
    chans = []
    for ch in range(nch):
        ar = _ar2(rng.standard_normal(n) * 0.02, 1.6, -0.7) * 0.3
        wn = rng.standard_normal(n) * 0.002
        chans.append(gains[ch % 2] * tone + ar + wn)
    x = np.stack(chans)
    x = x / np.max(np.abs(x)) * 0.7
    fs = float(1 << (bits - 1))
    q = np.rint(x * fs).astype(np.int64)
    q = np.clip(q, -(1 << (bits - 1)), (1 << (bits - 1)) - 1)
    if sparse_bits and sparse_bits < bits:
        sh = bits - sparse_bits
        q = (q >> sh) << sh
    return q.astype(np.int32)
