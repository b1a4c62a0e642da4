"""Pin the CPU restatement (oracle/sac_oracle.cpp) against golden vectors produced by the
genuine reference (tests/golden/make_golden.py -> oracle/_ref).  No GPU, no /root/reference."""
import zlib

import numpy as np
import pytest

from golden_cases import FRAMESIZE, chain_cases, frame_cases, trace_cases, trace_cases_r2
from oracle_api import center_frame


def test_profile_and_tables(orc, golden):
    assert np.array_equal(orc.profile(), golden["profile"])
    fwd, inv = orc.domain_tables()
    assert zlib.crc32(fwd.tobytes()) == int(golden["domain_crc"][0])
    assert zlib.crc32(inv.tobytes()) == int(golden["domain_crc"][1])
    assert np.array_equal(fwd[::257], golden["domain_fwd_sample"])


@pytest.mark.parametrize("name", list(trace_cases(np.zeros((58, 3), np.float32)).keys()))
def test_predictor_trace_bit_exact(orc, golden, name):
    raw = golden[f"trace/{name}/raw"]
    coefs = golden[f"trace/{name}/coefs"]
    _, _, opt, start, n = trace_cases(golden["profile"])[name]
    smp, stats = center_frame(raw)
    pd, plpc, plms, err = orc.predict_trace(smp, stats, coefs, start, n, opt)
    # integer residuals: bit-exact.  fp64 intermediates: bit-exact too (the restatement spells
    # out every fused multiply-add of the reference binary).
    assert np.array_equal(err, golden[f"trace/{name}/err"])
    assert np.array_equal(plpc.view(np.uint64), golden[f"trace/{name}/plpc"].view(np.uint64))
    assert np.array_equal(plms.view(np.uint64), golden[f"trace/{name}/plms"].view(np.uint64))
    assert np.array_equal(pd.view(np.uint64), golden[f"trace/{name}/pd"].view(np.uint64))
    # predict_frame (no trace) gives the same residuals
    err2, _ = orc.predict_frame(smp, stats, coefs, start, n, opt)
    assert np.array_equal(err2, err)


@pytest.mark.parametrize("name", list(trace_cases_r2(np.zeros((58, 3), np.float32)).keys()))
def test_predictor_trace_r2_bit_exact(orc, golden_r2, name):
    """The oracle against the round-2 reference traces (long / tiny / ragged NLMS stages, k = 1)."""
    raw = golden_r2[f"trace/{name}/raw"].astype(np.int32)
    coefs = golden_r2[f"trace/{name}/coefs"]
    smp, stats = center_frame(raw)
    n = raw.shape[1]
    pd, plpc, plms, err = orc.predict_trace(smp, stats, coefs, 0, n, 0)
    assert np.array_equal(err, golden_r2[f"trace/{name}/err"])
    assert np.array_equal(plpc.view(np.uint64), golden_r2[f"trace/{name}/plpc"].view(np.uint64))
    assert np.array_equal(plms.view(np.uint64), golden_r2[f"trace/{name}/plms"].view(np.uint64))


@pytest.mark.parametrize("name", list(frame_cases().keys()))
def test_frame_record_byte_exact_and_roundtrip(orc, golden, name):
    raw = golden[f"frame/{name}/raw"]
    cfg = frame_cases()[name][1]
    r = orc.encode_frame(raw, cfg, FRAMESIZE, trace=True)
    want = golden[f"frame/{name}/record"].tobytes()
    assert r["record"] == want
    assert np.array_equal(r["info"], golden[f"frame/{name}/info"])
    assert np.array_equal(r["profile"], golden[f"frame/{name}/profile"])
    if cfg.optimize:
        assert np.array_equal(r["trace_cost"], golden[f"frame/{name}/trace_cost"])
        assert np.array_equal(r["trace_coefs"], golden[f"frame/{name}/trace_coefs"])
    dec, _ = orc.decode_frame(want, raw.shape[0], FRAMESIZE)
    assert np.array_equal(dec, raw)


def test_mapped_cases_are_really_mapped(golden):
    assert golden["frame/sparse16_normal/info"][0, 1] == 1
    assert golden["frame/s16_normal/info"][0, 1] == 0


def test_coder_trace(orc, golden):
    u = golden["coder/s2u"]
    mb, cnt = [int(v) for v in golden["coder/maxbpn"]]
    c, p1, bits = orc.bitplane_trace(u, mb, 20000)
    assert c == cnt == u.size * (mb + 1)
    assert np.array_equal(p1, golden["coder/p1"])
    assert np.array_equal(bits, golden["coder/bits"])
    data = orc.bitplane_encode(u, mb)
    assert data == golden["coder/bytes"].tobytes()
    e = orc.bitplane_decode(data, u.size, mb)
    s2u = np.where(e < 0, -2 * e, np.where(e > 0, 2 * e - 1, 0))
    assert np.array_equal(s2u, u)
    # the range coder alone, fed the golden decisions for the first 20000 symbols
    rc = orc.rangecoder_encode(golden["coder/p1"], golden["coder/bits"])
    assert len(rc) > 5


def test_costs(orc, golden):
    e = golden["cost/err"]
    got = np.array([orc.cost(k, e) for k in range(5)])
    assert np.array_equal(got, golden["cost/values"])
    assert orc.cost(2, np.zeros(0, np.int32)) == 0.0


def test_rng_and_search(orc, golden):
    vals = orc.rng(golden["rng/kinds"], np.full(1000, 55.0))
    assert np.array_equal(vals, golden["rng/values"])
    assert np.array_equal(orc.gen_norm(0.3, 0.0, 1.0, 0.2, 300), golden["rng/gen_norm"])
    nd = 12
    lo = np.zeros(nd); hi = np.arange(1, nd + 1) * 1.0
    for nt in (0, 4):
        best, xb, tc = orc.dds_quadratic(lo, hi, hi * 0.5, hi * 0.25, 120, nt, 0.2)
        assert np.array_equal(xb, golden[f"dds/q{nt}/xbest"])
        assert np.allclose(tc, golden[f"dds/q{nt}/trace"], rtol=1e-13, atol=0)
        assert best == tc.min()


def test_remap(orc, golden):
    raw = golden["frame/sparse16_normal/raw"][0]
    smp, stats = center_frame(raw[None, :])
    err, pred = orc.predict_frame(smp, stats, golden["profile"][:, 2].copy(), 0, raw.size, 0)
    r, s2u_map, mb, ul, uh = orc.remap(raw, pred[0], err[0])
    assert r == float(golden["remap/ratio"][0]) and r > 1.05
    assert np.array_equal(s2u_map, golden["remap/s2u_map"])
    assert mb == int(golden["remap/maxbpn"][0])
    assert orc.mapencode(ul, uh) == golden["remap/mapbytes"].tobytes()


def test_analyse(orc):
    x = np.array([5, -7, 9, 100, -3], np.int32)
    assert orc.analyse(x).tolist() == [20, -7, 100]  # floor(104/5)=20
    x = np.array([-5, -6], np.int32)
    assert orc.analyse(x).tolist() == [-6, -6, -5]  # floor(-5.5)


# ---- adaptive sub-frame split (Codec::Analyse / PushState / SparsePCM), SURVEY section 8(f) rank 1
def _subgolden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subframes_golden.npz"))


@pytest.mark.parametrize("name", ["dense_only", "sparse_tail", "alternating", "short_last_block", "min_two_blocks", "all_sparse"])
def test_subframe_plan_matches_reference(orc, name):
    g = _subgolden()
    pcm = g[f"{name}/pcm"].astype(np.int32)
    blk, min_len = (int(v) for v in g[f"{name}/args"])
    want = [tuple(int(v) for v in row) for row in g[f"{name}/subframes"]]
    assert orc.plan_subframes(pcm, blk, min_len) == want
    for ch in range(pcm.shape[0]):
        used, cost = orc.sparse_cost(pcm[ch, :blk])
        assert (used, cost) == tuple(g[f"{name}/cost_block0"][ch])        # bit-exact doubles


@pytest.mark.parametrize("name", list(chain_cases().keys()))
def test_warm_start_chain_vs_golden(orc, golden_r3, name):
    """reset=0 (the reference's default): frame f+1's search starts from frame f's optimum (libsac.cpp:461-466)."""
    frames, cfg = chain_cases()[name]
    prof = None
    for f in range(len(frames)):
        raw = golden_r3[f"chain/{name}/{f}/raw"].astype(np.int32)
        r = orc.encode_frame(raw, cfg, FRAMESIZE, profile=prof)
        prof = r["profile"]
        assert r["record"] == golden_r3[f"chain/{name}/{f}/record"].tobytes(), (name, f)
        assert np.array_equal(prof, golden_r3[f"chain/{name}/{f}/profile"])


@pytest.mark.parametrize("name", list(__import__("golden_cases").wide_cases().keys()))
def test_wide_material_records_vs_golden(orc, golden_r4, name):
    """24-bit material (--sparse-pcm=0): the oracle's records, profiles and search costs equal the genuine reference's."""
    from golden_cases import wide_cases
    raw = golden_r4[f"wide/{name}/raw"]
    cfg = wide_cases()[name][1]
    assert np.array_equal(raw, wide_cases()[name][0])
    r = orc.encode_frame(raw, cfg, FRAMESIZE, trace=True)
    assert r["record"] == golden_r4[f"wide/{name}/record"].tobytes()
    assert np.array_equal(r["profile"], golden_r4[f"wide/{name}/profile"])
    if cfg.optimize:
        assert np.array_equal(r["trace_cost"][: cfg.maxnfunc], golden_r4[f"wide/{name}/trace_cost"][: cfg.maxnfunc])
    dec, _ = orc.decode_frame(r["record"], raw.shape[0], FRAMESIZE)
    assert np.array_equal(dec, raw)


@pytest.mark.parametrize("name", list(__import__("golden_cases").search_cases().keys()))
def test_de_and_cma_records_vs_golden(orc, golden_r4, name):
    """--opt-cfg=de / cma: the oracle's own restatement of OptDE::run / OptCMA::run (sac_oracle.cpp) reproduces the records,
    chosen profiles and every search cost of oracle/_ref (DriverDE / DriverCMA around the genuine Opt, Cholesky, SSC1)."""
    from golden_cases import search_cases
    raw, cfg, search = search_cases()[name]
    r = orc.encode_frame(raw, cfg, FRAMESIZE, trace=True, search=search)
    assert r["record"] == golden_r4[f"search/{name}/record"].tobytes()
    assert np.array_equal(r["profile"], golden_r4[f"search/{name}/profile"])
    want = golden_r4[f"search/{name}/trace_cost"]
    assert len(r["trace_cost"]) == len(want) and np.array_equal(r["trace_cost"], want)
    orc.lib.orc_set_search_method(0)
