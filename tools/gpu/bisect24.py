"""Round 4, GPU bisect of the 24-bit record mismatch.  Run on the GPU box:
  python tools/gpu/bisect24.py           (wave-parallel decision chain, the product's)
  SACAMD_CODER_SERIAL=1 python tools/gpu/bisect24.py   (lane-0 chain, the body the CPU emulation runs)
Prints one line per probe: (a) device libm / PredictLaplace vs the host libm, (b) residuals vs the oracle,
(c) coder bytes vs the oracle for residuals of growing width."""
import math, os, struct, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sac_amd.api as api
from oracle_api import Checker, center_frame
from golden_cases import wide_cases, FRAMESIZE

orc = Checker("orc")
print("serial_chain", os.environ.get("SACAMD_CODER_SERIAL", "0"))


def host_laplace(avg, bpn):
    p_l = 0.0
    if avg > 0:
        theta = math.exp(-1.0 / avg)
        try:
            pw = math.pow(theta, float(1 << bpn))
        except OverflowError:
            pw = math.inf
        p_l = 1.0 - 1.0 / (1 + pw)
    return min(max(int(math.floor(p_l * 32768 + 0.5)), 1), 32767)


ctx = api.Context(2, FRAMESIZE, 1)
# ---- (a) libm taps
rng = np.random.default_rng(1)
x = np.concatenate([-1.0 / rng.integers(1, 1 << 25, 200000), -rng.uniform(0, 800, 100000), rng.uniform(-1e-9, 1e-9, 1000), -10.0 ** rng.uniform(-12, 3, 100000)])
got = ctx.debug_libm(0, x)
want = np.array([math.exp(v) for v in x])
bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
print("exp mismatches", bad.size, "of", x.size, [(x[i], got[i], want[i]) for i in bad[:5]])
avg = rng.integers(1, 1 << 25, 300000)
th = np.array([math.exp(-1.0 / a) for a in avg])
yy = 2.0 ** rng.integers(0, 25, avg.size)
got = ctx.debug_libm(1, th, yy)
want = np.array([math.pow(a, b) for a, b in zip(th, yy)])
bad = np.nonzero(got.view(np.uint64) != want.view(np.uint64))[0]
print("pow mismatches", bad.size, "of", th.size, [(th[i], yy[i], got[i], want[i]) for i in bad[:5]])
avg = np.concatenate([rng.integers(0, 1 << 25, 300000), np.arange(0, 4096), (1 << np.arange(0, 25)), (1 << np.arange(1, 25)) - 1])
bp = rng.integers(0, 25, avg.size)
got = ctx.debug_libm(2, avg.astype(np.float64), bp.astype(np.float64))
want = np.array([host_laplace(int(a), int(b)) for a, b in zip(avg, bp)], np.float64)
bad = np.nonzero(got != want)[0]
print("laplace mismatches", bad.size, "of", avg.size, [(int(avg[i]), int(bp[i]), got[i], want[i]) for i in bad[:8]])

# ---- (b) residuals of the failing frame
raw, cfg = wide_cases()["s24_normal"]
smp, stats = center_frame(raw)
prof = api.default_profile()[:, 2].copy()
oerr, opred = orc.predict_frame(smp, stats, prof, 0, raw.shape[1], 0)
ctx.upload_i32([raw], FRAMESIZE)
g = api.make_cfg("normal", sparse_pcm=0)
ctx.analyse(g)
plpc, psum, err, pred = ctx.debug_predict(0, prof, 0, raw.shape[1], 0, 4)
print("residuals equal", np.array_equal(err, oerr), "pred equal", np.array_equal(pred, opred))

# ---- (c) coder on residual vectors of growing width
def s2u(e):
    e = e.astype(np.int64)
    return np.where(e < 0, -2 * e, np.where(e > 0, 2 * e - 1, 0)).astype(np.int32)

def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return -1 if len(a) == len(b) else n

for ch in range(raw.shape[0]):
    u = s2u(oerr[ch]); mb = int(np.floor(np.log2(max(int(u.max()), 1))))
    a = ctx.debug_bitplane(u, mb); b = orc.bitplane_encode(u, mb)
    print(f"s24_normal ch{ch} maxbpn {mb}: len {len(a)} vs {len(b)} first diff {first_diff(a, b)}")
    for sh in (1, 2, 3, 4, 5, 6, 7, 8):
        us = (u >> sh).astype(np.int32); mbs = int(np.floor(np.log2(max(int(us.max()), 1))))
        a = ctx.debug_bitplane(us, mbs); b = orc.bitplane_encode(us, mbs)
        print(f"   >> {sh} maxbpn {mbs}: len {len(a)} vs {len(b)} first diff {first_diff(a, b)}")
    for nn in (64, 128, 256, 1000):
        a = ctx.debug_bitplane(u[:nn], mb); b = orc.bitplane_encode(u[:nn], mb)
        print(f"   first {nn} samples: len {len(a)} vs {len(b)} first diff {first_diff(a, b)}")
rng = np.random.default_rng(7)
for scale, n in ((3e5, 3000), (2e6, 2500), (8e4, 1000), (2e4, 1000)):
    e = np.clip(np.rint(rng.laplace(size=n) * scale).astype(np.int64), -(1 << 23), (1 << 23) - 1).astype(np.int32)
    u = s2u(e); mb = int(np.floor(np.log2(max(int(u.max()), 1))))
    a = ctx.debug_bitplane(u, mb); b = orc.bitplane_encode(u, mb)
    print(f"laplace noise scale {scale:g} maxbpn {mb}: len {len(a)} vs {len(b)} first diff {first_diff(a, b)}")
# whole record
recs, profs = ctx.encode_frames(g)
gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_golden_r4.npz"))["wide/s24_normal/record"].tobytes()
print("record equal", recs[0] == gold, "first diff", first_diff(recs[0], gold))
ctx.close()
