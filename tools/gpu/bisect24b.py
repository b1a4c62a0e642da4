"""Round 4: which predictor stage diverges on 24-bit input (device vs oracle trace)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sac_amd.api as api
from oracle_api import Checker, center_frame
from golden_cases import wide_cases, FRAMESIZE
from sac_amd.synth import synth_pcm
orc = Checker("orc")
prof = api.default_profile()[:, 2].copy()

def probe(name, raw, opt=0):
    smp, stats = center_frame(raw)
    n = raw.shape[1]
    pd, oplpc, oplms, oerr = orc.predict_trace(smp, stats, prof, 0, n, opt)
    ctx = api.Context(raw.shape[0], FRAMESIZE, 1)
    ctx.upload_i32([raw], FRAMESIZE)
    ctx.analyse(api.make_cfg("normal", sparse_pcm=0))
    st = ctx.stats()
    plpc, psum, err, pred = ctx.debug_predict(0, prof, 0, n, opt, 4)
    ctx.close()
    print(name, "opt", opt, "stats", st.reshape(-1).tolist(), "oracle stats", np.asarray(stats).reshape(-1).tolist())
    for ch in range(raw.shape[0]):
        a = np.nonzero(plpc[ch].view(np.uint64) != oplpc[ch].view(np.uint64))[0]
        osum = oplpc[ch] + oplms[ch]
        b = np.nonzero(psum[ch].view(np.uint64) != osum.view(np.uint64))[0]
        c = np.nonzero(err[ch] != oerr[ch])[0]
        print(f"  ch{ch}: plpc mismatches {a.size} first {a[:3].tolist()}  psum mismatches {b.size} first {b[:3].tolist()}  err mismatches {c.size} first {c[:3].tolist()}")
        for idx, nm, x, y in ((a, "plpc", plpc[ch], oplpc[ch]), (b, "psum", psum[ch], osum), (c, "err", err[ch], oerr[ch])):
            if idx.size:
                i = int(idx[0]); print(f"     {nm}[{i}] gpu {x[i]!r} oracle {y[i]!r}  raw {raw[ch, max(0, i - 2): i + 1].tolist()} pd {pd[ch][i]!r}")

raw, _ = wide_cases()["s24_normal"]
probe("s24_normal", raw)
probe("s24_normal", raw, 1)
probe("s24 mono ch0", raw[:1])
for sh in (2, 4, 6, 8):
    probe(f"s24 >> {sh}", (raw >> sh).astype(np.int32))
probe("s16", synth_pcm(5000, 2, 91, 8000))
