"""CPU-side checks of the PRODUCT code (no GPU): the C-ABI library loads and exports every symbol
declared in include/sac_amd.h; the kernel bodies (sac_amd/csrc/pred_*.h, coder.h), run lane by
lane through the emulation executor, reproduce the golden vectors; the host DDS driver and the
device libm port agree with the reference behaviour; the multi-rank record gather works (gloo)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from golden_cases import trace_cases, trace_cases_r2
from oracle_api import _vp, center_frame, frame_cfg
from sac_amd.synth import synth_pcm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so"))
    lib.emu_dds_quadratic.restype = ctypes.c_double
    lib.emu_libm_mismatches.restype = ctypes.c_long
    return lib


def test_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sac_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(sacamd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    so = os.path.join(ROOT, "sac_amd", "libsac_amd.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "sac_amd", "csrc")])
    lib = ctypes.CDLL(so)           # loads without a GPU; no compute call is made here
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    import sac_amd.api as api
    assert sorted(api.ABI_SYMBOLS) == declared
    assert lib.sacamd_abi_version() == api.ABI_VERSION == 7
    # without a GPU the context constructor must fail loudly (no CPU fallback)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(api.SacAmdError):
            api.Context(2, 1000, 1)


def test_inline_asm_dpp_instructions_have_no_hazard_producers():
    """The backward solve of k_ols_grid issues v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast through inline assembly (simt.h), which
    LLVM's hazard recogniser does not inspect: the built library's ISA must not feed a DPP source from a VALU write (2 wait states)
    or follow an EXEC write (5) too closely.  tools/check_dpp_hazard.py disassembles libsac_amd.so's gfx950 code objects."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "tools", "check_dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    n, bad = mod.check(os.path.join(ROOT, "sac_amd", "libsac_amd.so"))
    assert n > 1000, n                       # the grid kernels' backward solves are there
    assert not bad, bad[:5]


def test_cascade_launch_plan_keeps_small_launches_off_the_busy_streams():
    """sacamd_plan_cascade_streams (launch_plan.h), the planner run_predict uses.  The case is the late cascade group of the
    final pass of the 1536-frame step (profiles/r06/launch_trace_1536_before_streams.txt): two throughput-bound launches and
    seven launches of 1-39 items on five streams.  Balanced by taps, four of the small ones had queued behind one another."""
    import sac_amd.api as api
    n = 882000.0
    items = [879, 590, 12, 39, 5, 4, 1, 1, 1]
    taps = [7000, 4000, 7500, 1500, 7500, 7500, 7600, 7400, 7300]
    work = [i * t * n for i, t in zip(items, taps)]
    order, stream = api.plan_cascade_streams([1] * 9, work, [[8, 9, 10, 11, 0, 1, 2], [3, 4, 5, 6, 7]], 1.5e6 * n)
    assert sorted(order.tolist()) == list(range(9)) and order[0] == 0 and order[1] == 1            # longest first
    assert set(stream.tolist()) <= {3, 4, 5, 6, 7}                                                 # the group's own pool
    big = {int(stream[0]), int(stream[1])}
    assert len(big) == 2
    small = [int(stream[q]) for q in range(2, 9)]
    assert not (set(small) & big)                                                                  # never behind a launch that ends with the chip's drain
    assert max(small.count(s) for s in set(small)) == 3 and len(set(small)) == 3                   # 7 over 3 streams: 3 + 2 + 2
    # every stream of the pool busy (a search generation): small launches queue behind the SMALLEST throughput-bound launch
    work2 = [1e300, 9e6 * n, 8e6 * n, 7e6 * n, 2e6 * n, 1e5 * n, 2e5 * n]
    order2, stream2 = api.plan_cascade_streams([0] * 7, work2, [[3, 4, 5, 6, 7]], 5e5 * n)
    assert order2[0] == 0                                                                          # the whole-CU layout is issued first
    assert stream2[5] == stream2[6] == stream2[4] and len(set(stream2[:5].tolist())) == 5
    # groups are planned independently and in group order; a group without streams is an argument error
    order3, stream3 = api.plan_cascade_streams([1, 0, 1, 0], [5e6, 1e6, 9e6, 3e6], [[0, 1], [2, 3]], 1.0)
    assert order3.tolist() == [3, 1, 2, 0] and stream3.tolist() == [3, 1, 2, 0]
    with pytest.raises(api.SacAmdError):
        api.plan_cascade_streams([1], [1.0], [[0], []], 1.0)
    o0, s0 = api.plan_cascade_streams([], [], [[0]], 1.0)
    assert len(o0) == 0 and len(s0) == 0


def test_default_profile_matches_reference(golden):
    import sac_amd.api as api
    assert np.array_equal(api.default_profile(), golden["profile"])


@pytest.mark.parametrize("name", list(trace_cases(np.zeros((58, 3), np.float32)).keys()))
def test_kernel_bodies_emulated_vs_golden(emu, golden, name):
    raw = golden[f"trace/{name}/raw"]
    coefs = np.ascontiguousarray(golden[f"trace/{name}/coefs"], np.float32)
    _, _, opt, start, n = trace_cases(golden["profile"])[name]
    smp, stats = center_frame(raw)
    nch, total = smp.shape
    plpc = np.zeros((nch, n)); psum = np.zeros((nch, n))
    err = np.zeros((nch, n), np.int32); pred = np.zeros((nch, n), np.int32)
    rc = emu.emu_predict(nch, total, _vp(np.ascontiguousarray(smp, np.int32)), _vp(np.ascontiguousarray(stats, np.int32)),
                         _vp(coefs), start, n, int(opt), 4, _vp(plpc), _vp(psum), _vp(err), _vp(pred))
    assert rc == 0
    assert np.array_equal(err, golden[f"trace/{name}/err"])
    rl = golden[f"trace/{name}/plpc"]
    assert np.array_equal(plpc.view(np.uint64), rl.view(np.uint64))      # OLS stage: bit-exact
    ps = rl + golden[f"trace/{name}/plms"]
    if opt:      # search evaluations (k = optk): free-order NLMS sums, tolerance stated in DESIGN.md
        assert np.max(np.abs(psum - ps) / (np.abs(ps) + 1.0)) < 1e-9
    else:        # final pass (k = 1, what the decoder recomputes): slmath::dot order -> bit-exact
        assert np.array_equal(psum.view(np.uint64), ps.view(np.uint64))


@pytest.mark.parametrize("name", list(trace_cases_r2(np.zeros((58, 3), np.float32)).keys()))
def test_canonical_cascade_layouts_emulated_vs_golden(emu, golden_r2, name):
    """Final-pass cascade (slmath::dot / calc_s2pow order) in every layout class -- 256 lanes, 512 lanes, the
    profile maximum -- and with every kind of transform_reduce tail: p_lpc + p_lms bit-identical to the reference."""
    raw = golden_r2[f"trace/{name}/raw"].astype(np.int32)
    coefs = np.ascontiguousarray(golden_r2[f"trace/{name}/coefs"], np.float32)
    smp, stats = center_frame(raw)
    nch, n = smp.shape
    plpc = np.zeros((nch, n)); psum = np.zeros((nch, n))
    err = np.zeros((nch, n), np.int32); pred = np.zeros((nch, n), np.int32)
    rc = emu.emu_predict(nch, n, _vp(np.ascontiguousarray(smp, np.int32)), _vp(np.ascontiguousarray(stats, np.int32)),
                         _vp(coefs), 0, n, 0, 4, _vp(plpc), _vp(psum), _vp(err), _vp(pred))
    assert rc == 0
    rl = golden_r2[f"trace/{name}/plpc"]
    assert np.array_equal(plpc.view(np.uint64), rl.view(np.uint64))
    ps = rl + golden_r2[f"trace/{name}/plms"]
    assert np.array_equal(psum.view(np.uint64), ps.view(np.uint64))
    assert np.array_equal(err, golden_r2[f"trace/{name}/err"])


def test_coder_body_emulated_vs_golden(emu, orc, golden):
    fwd, inv = orc.domain_tables()
    u = np.ascontiguousarray(golden["coder/s2u"], np.int32)
    mb = int(golden["coder/maxbpn"][0])
    out = np.zeros(u.size * 4 + 70000, np.uint8)
    n = emu.emu_bitplane(_vp(u), u.size, mb, None, _vp(fwd), _vp(inv), _vp(out), out.size)
    assert out[:n].tobytes() == golden["coder/bytes"].tobytes()
    # mapped variant: MapEncoder prefix + remapped residual == payload of the golden sparse frame
    raw = golden["frame/sparse16_normal/raw"][0]
    smp, stats = center_frame(raw[None, :])
    err, pred = orc.predict_frame(smp, stats, golden["profile"][:, 2].copy(), 0, raw.size, 0)
    r, s2m, mbm, ul, uh = orc.remap(raw, pred[0], err[0])
    used = np.ascontiguousarray(np.concatenate([ul, uh]), np.uint8)
    n = emu.emu_bitplane(_vp(np.ascontiguousarray(s2m, np.int32)), s2m.size, mbm, _vp(used), _vp(fwd), _vp(inv), _vp(out), out.size)
    assert out[:n].tobytes() == golden["frame/sparse16_normal/record"].tobytes()[4 + 232 + 18:]


def test_coder_body_on_wide_material_vs_oracle(emu, orc):
    """Residuals wider than 16 bits: planes above 17 and avg_sum beyond the 2^17-entry table take PredictLaplace evaluated in
    the kernel (laplace_direct: the glibc exp / pow ports) -- same value as the host libm expression for every (avg, plane),
    same bytes as the oracle's coder, and the decoder body inverts them."""
    emu.emu_laplace_mismatches.restype = ctypes.c_long
    assert emu.emu_laplace_mismatches(1_000_000, 5) == 0
    fwd, inv = orc.domain_tables()
    rng = np.random.default_rng(7)
    for scale, n in ((3e5, 3000), (2e6, 2500), (8e4, 1000)):
        e = np.clip(np.rint(rng.laplace(size=n) * scale).astype(np.int64), -(1 << 23), (1 << 23) - 1).astype(np.int32)
        u = np.where(e < 0, -2 * e, np.where(e > 0, 2 * e - 1, 0)).astype(np.int32)
        mb = int(np.floor(np.log2(max(int(u.max()), 1))))
        out = np.zeros(u.size * 4 + 70000, np.uint8)
        ln = emu.emu_bitplane(_vp(u), u.size, mb, None, _vp(fwd), _vp(inv), _vp(out), out.size)
        want = orc.bitplane_encode(u, mb)
        assert out[:ln].tobytes() == want, (scale, mb)
        dec = np.zeros(n, np.int32); pl = np.frombuffer(want, np.uint8).copy()
        emu.emu_bitplane_decode(_vp(pl), pl.size, n, mb, None, _vp(fwd), _vp(inv), _vp(dec))
        assert np.array_equal(dec, u)


def test_prediction_conversion_follows_the_x86_reference(emu, orc):
    """`(int32_t)std::round(pd)` (libsac.cpp:106): outside the int32 range the reference's x86-64 build yields INT_MIN (cvttsd2si),
    which the clamp turns into the frame minimum; gfx950's conversion saturates instead, so the product spells the x86 result
    out (canon.h: cvt_i32_x86).  24-bit material reaches |pd| > 2^31 in the first samples of a frame: the kernel body on such a
    frame equals the oracle, whose residual there is val - minval."""
    emu.emu_cvt_i32.argtypes = [ctypes.c_double]
    lo = -(1 << 31)
    for v, want in ((2250366745.0, lo), (-3e9, lo), (float("nan"), lo), (float("inf"), lo), (2147483647.0, 2147483647), (2147483648.0, lo),
                    (-2147483648.0, lo), (-2147483649.0, lo), (12345.0, 12345), (-7.0, -7)):
        assert emu.emu_cvt_i32(v) == want, v
    from golden_cases import wide_cases
    raw, _ = wide_cases()["s24_normal"]
    smp, stats = center_frame(raw)
    prof = orc.profile()[:, 2].copy()
    n = raw.shape[1]
    pd, _, _, oerr = orc.predict_trace(smp, stats, prof, 0, n, 0)
    over = np.abs(pd) >= 2.0 ** 31
    assert over.any()                                   # the case exists in this frame (ch0, samples 19..21)
    assert np.array_equal(oerr[over], (smp - np.asarray(stats).reshape(-1, 3)[:, :1])[over])      # INT_MIN clamped to minval
    plpc = np.zeros((2, n)); psum = np.zeros((2, n)); err = np.zeros((2, n), np.int32); pred = np.zeros((2, n), np.int32)
    rc = emu.emu_predict(2, n, _vp(np.ascontiguousarray(smp, np.int32)), _vp(np.ascontiguousarray(stats, np.int32)),
                         _vp(np.ascontiguousarray(prof, np.float32)), 0, n, 0, 4, _vp(plpc), _vp(psum), _vp(err), _vp(pred))
    assert rc == 0 and np.array_equal(err, oerr)


@pytest.mark.parametrize("case", [(1, 17, 0, 0), (1, 24, 0, 1), (1, 25, 0, 1), (1, 32, 0, 0), (1, 32, 1, 1), (1, 32, 8, 0), (1, 32, 9, 1), (1, 32, 16, 0), (1, 32, 17, 1),
                                  (1, 32, 24, 0), (1, 32, 25, 1), (1, 32, 32, 0), (1, 32, 32, 1), (2, 20, 13, 0), (2, 32, 29, 1), (2, 31, 30, 0)])
def test_grid_ols_kernel_body_vs_oracle(emu, orc, case):
    """pred_ols_grid.h (matrix 2D-cyclic over the lanes of one wave, backward solve as one row-broadcast pass over the linear
    stream): p_lpc bit-identical to the oracle for every block count NB = 3 .. 8 incl. the class boundaries (17, 24 | 25, 32 | 33,
    40 | 41 ... 64 taps), mono (nA + nM0) and stereo slot 1 (nB + nS0 + |nS1|, ch_ref swap), k = 1 (final pass) and k = optk."""
    nch, a, b, opt = case
    rng = np.random.default_rng(1000 * a + 10 * b + opt)
    n = 700
    raw = synth_pcm(n, nch, seed=77 + a + b, rate=8000)
    smp, stats = center_frame(raw)
    g = orc.profile()[:, 2].copy()
    if nch == 1:
        g[24] = a; g[9] = b
    else:
        g[24] = 12; g[9] = 0
        g[25] = a; g[26] = b // 2; g[27] = (b - b // 2) * (-1 if opt else 1)       # slot 1: nB + nS0 + |nS1| taps; nS1 < 0: ch_ref = 1
    g[0] = rng.uniform(0.99, 0.9999); g[12] = rng.uniform(0.99, 0.9999); g[1] = rng.uniform(1, 100); g[13] = rng.uniform(1, 100)
    pd, plpc, plms, oerr = orc.predict_trace(smp, stats, g, 0, n, opt)
    pl = np.zeros((nch, n)); ps = np.zeros((nch, n)); err = np.zeros((nch, n), np.int32); pred = np.zeros((nch, n), np.int32)
    emu.emu_set_ols_grid(1)
    rc = emu.emu_predict(nch, n, _vp(np.ascontiguousarray(smp, np.int32)), _vp(np.ascontiguousarray(stats, np.int32)), _vp(np.ascontiguousarray(g, np.float32)),
                         0, n, opt, 4, _vp(pl), _vp(ps), _vp(err), _vp(pred))
    assert rc == 0
    assert np.array_equal(pl.view(np.uint64), plpc.view(np.uint64))
    if not opt:
        assert np.array_equal(err, oerr)


def test_lane_map_layout_choice_respects_the_lds(emu):
    """canon3_class_for (pred_lms.h): a lane-map layout is taken only if the item's rings + lane-major mutab block fit the 160 KB of
    one CU's LDS (round 5: the 512-lane layout used to be chosen by its chains alone and asked for more); what no lane-map layout
    can take goes to the systolic layout (9).  Property over the whole profile box + the cases that failed on the GPU."""
    emu.emu_canon_lds_bytes.restype = ctypes.c_long
    cls = lambda vn: emu.emu_canon_class((ctypes.c_int * 4)(*vn))
    lds = lambda vn, c: emu.emu_canon_lds_bytes((ctypes.c_int * 4)(*vn), c)
    rng = np.random.default_rng(11)
    seen = set()
    for _ in range(4000):
        vn = (int(rng.integers(256, 8193)), int(rng.integers(32, 4097)), int(rng.integers(4, 2049)), int(rng.integers(2, 1025)))
        c = cls(vn); seen.add(c)
        assert c in (9, 10, 11, 12, 13)
        if c != 9:
            assert lds(vn, c) <= 160 * 1024, (vn, c)
    assert seen == {9, 10, 11, 12, 13}
    assert cls((1280, 256, 32, 4)) == 10 and cls((8192, 4096, 2048, 1024)) == 9
    big = [vn for vn in ((6600, 300, 40, 8), (6900, 64, 8, 8), (7200, 32, 4, 2)) if cls(vn) == 13]
    assert big and all(lds(vn, 13) <= 160 * 1024 for vn in big)


def test_host_dds_driver_matches_reference_search(emu, golden):
    nd = 12
    lo = np.zeros(nd); hi = np.arange(1, nd + 1) * 1.0
    xs = hi * 0.5; c = hi * 0.25
    for nt in (0, 4):
        xb = np.zeros(nd); tc = np.zeros(120)
        emu.emu_dds_quadratic(nd, _vp(lo), _vp(hi), _vp(xs), _vp(c), 120, nt, ctypes.c_double(0.2), _vp(xb), _vp(tc))
        assert np.array_equal(xb, golden[f"dds/q{nt}/xbest"])
        assert np.allclose(tc, golden[f"dds/q{nt}/trace"], rtol=1e-13, atol=0)


@pytest.mark.parametrize("name", list(__import__("golden_cases").search_quadratic_cases().keys()))
def test_host_de_cma_searchers_match_reference(emu, golden_r4, name):
    """search_host.h (FrameSearchDE / FrameSearchCMA, product host code) vs opt/de.cpp / opt/cma.cpp as oracle/_ref runs
    them: every evaluated point's cost and the best point bit-identical (the test function has no multiply-add, so it
    is the same with and without FMA contraction)."""
    from golden_cases import search_quadratic_cases, search_quadratic_inputs
    search, ndim, nmax, sigma, seed = search_quadratic_cases()[name]
    lo, hi, xs, cen = search_quadratic_inputs(ndim, seed)
    xb = np.zeros(ndim); tc = np.zeros(nmax + 64); ne = ctypes.c_int(0)
    emu.emu_search_quadratic.restype = ctypes.c_double
    emu.emu_search_quadratic(search, ndim, _vp(lo), _vp(hi), _vp(xs), _vp(cen), nmax, 1, ctypes.c_double(sigma), _vp(xb), _vp(tc), ctypes.byref(ne))
    want = golden_r4[f"quad/{name}/trace"]
    assert ne.value == len(want)
    assert np.array_equal(tc[: ne.value], want)
    assert np.array_equal(xb, golden_r4[f"quad/{name}/xbest"])


def test_device_libm_port_is_bit_exact_with_host_libm(emu):
    assert emu.emu_libm_mismatches(2_000_000, 3) == 0


GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
frames = bench.shard_frames(7, rank, world)
recs = [bytes([f]) * (10 + 3 * f) for f in frames]          # variable-length fake records
out = bench.gather_records(recs, rank, world, torch.device("cpu"))
if rank == 0:
    want = [bytes([f]) * (10 + 3 * f) for f in range(7)]
    assert out == want, (out, want)
else:
    assert out is None
# cost-based assignment (sacamd_assign_frames, longest first): ragged frame counts per rank, frame order restored
cost = [5.0, 1.0, 1.0, 1.0, 9.0, 2.0, 2.0, 0.5, 0.5]
mine = bench.shard_frames(len(cost), rank, world, cost=cost)
assert len(mine) == (3 if rank == 0 else 6), mine     # longest-first: rank 0 takes the 9 and two small frames
recs = [bytes([f]) * (3 + 2 * f) for f in mine]
out = bench.gather_records(recs, rank, world, torch.device("cpu"))
owners = [None] * len(cost)
import sac_amd.api as api
ow = api.assign_frames(cost, world)
if rank == 0:
    # rank 0 receives rank-major order; bench.py's own --scaling strong path (restore_frame_order, what its gather() calls)
    # puts the records back into frame order with the ownership sacamd_assign_frames gave every rank
    assert sorted(mine) == [f for f in range(len(cost)) if ow[f] == 0]
    assert bench.restore_frame_order(out, cost, world, len(cost)) == [bytes([f]) * (3 + 2 * f) for f in range(len(cost))]
else:
    assert out is None and sorted(mine) == [f for f in range(len(cost)) if ow[f] == 1]
# the library's own gather (sacamd_gather_records_via == the gather_core the RCCL communicator drives) over a gloo transport:
# cost-based ownership, ragged counts, records arrive on rank 0 in FRAME order; an empty record and an empty rank included
sys.path.insert(0, {tests!r})
from gloo_transport import make_transport
tr = make_transport(rank, world)
recs = [bytes([f]) * (3 + 2 * f) if f != 7 else b"" for f in mine]
out = api.gather_records_via(tr, mine, recs, len(cost))
if rank == 0:
    assert out == [bytes([f]) * (3 + 2 * f) if f != 7 else b"" for f in range(len(cost))], out
else:
    assert out is None
ids = list(range(5)) if rank == 1 else []                     # rank 0 owns nothing
out = api.gather_records_via(tr, ids, [bytes([9 - f]) * (1 + f) for f in ids], 5)
assert (out == [bytes([9 - f]) * (1 + f) for f in range(5)]) if rank == 0 else out is None
# ids that are not a partition (frame 2 on both ranks): every rank gets the same error, nobody hangs
try:
    api.gather_records_via(tr, [2, rank], [b"x", b"y"], 3)
    raise AssertionError("duplicate frame id accepted")
except api.SacAmdError:
    pass
# receive buffer too small on rank 0: its capacity travels in the first all-gather, EVERY rank returns the error before a
# payload moves (round 3: rank 0 alone, after serving the peers)
try:
    api.gather_records_via(tr, [rank], [b"z" * 100], 2, cap=150)
    raise AssertionError("too small a receive buffer accepted")
except api.SacAmdError:
    pass
# a rank whose own work failed joins the gather with nrec = -1: nobody waits for its records, every rank gets an error
lib = api.load_library()
import ctypes
import numpy as np
ids = np.array([rank], np.int32); blob = np.frombuffer(b"q" * 10, np.uint8).copy(); off = np.array([0, 10], np.int64)
out = np.zeros(64, np.uint8); out_off = np.zeros(3, np.int64)
rc = lib.sacamd_gather_records_via(ctypes.byref(tr), -1 if rank == 1 else 1, api._vp(ids), api._vp(blob), api._vp(off), 2,
                                   api._vp(out) if rank == 0 else None, ctypes.c_longlong(64 if rank == 0 else 0), api._vp(out_off) if rank == 0 else None)
assert rc == -6, rc        # SACAMD_ERR_COMM on EVERY rank, the reporting one included (include/sac_amd.h)
# bad arguments on rank 0 only (no out_off): both ranks return an error together
rc = lib.sacamd_gather_records_via(ctypes.byref(tr), 1, api._vp(ids), api._vp(blob), api._vp(off), 2, api._vp(out) if rank == 0 else None,
                                   ctypes.c_longlong(64 if rank == 0 else 0), None)
assert rc != 0, rc
# ... and the transport is still usable afterwards
out2 = api.gather_records_via(tr, [rank], [bytes([rank]) * 5], 2)
assert (out2 == [b"\x00" * 5, b"\x01" * 5]) if rank == 0 else out2 is None
if rank == 0:
    print("GATHER_OK")
dist.barrier(); dist.destroy_process_group()
"""


def test_record_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(GLOO_WORKER.format(root=ROOT, tests=os.path.join(ROOT, "tests")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29577", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout


def test_subframe_state_machine_host(orc):
    """sacamd_subframes_from_states (host part of sacamd_plan_subframes) against the oracle's planner,
    with the block states taken from the oracle's SparsePCM cost."""
    import sac_amd.api as api
    from golden_cases import subframe_cases
    for name, (pcm, blk, min_len) in subframe_cases().items():
        n = pcm.shape[1]
        lens = [min(blk, n - a) for a in range(0, n, blk)]
        states = []
        for b, ln in enumerate(lens):
            c = np.mean([orc.sparse_cost(pcm[ch, b * blk: b * blk + ln])[1] for ch in range(pcm.shape[0])])
            states.append(int(c > 1.35))
        assert api.subframes_from_states(states, lens, min_len) == orc.plan_subframes(pcm, blk, min_len), name


def test_wav_container_roundtrip():
    """WAV parse -> metadata pack -> unpack -> rebuilt WAV is byte-identical (incl. odd-sized and
    trailing chunks); header fields as Wav::ReadHeader reads them."""
    from sac_amd import container as C
    from sac_amd.synth import synth_pcm
    pcm = synth_pcm(1001, 2, 5, 8000)
    extra = [(0x5453494C, b"INFOISFT\x05\x00\x00\x00abcde"), (0x6B6E756A, b"xyz")]     # LIST (odd inner), 'junk' odd size
    blob = C.wav_bytes_from_pcm(pcm, 8000, 16, extra_chunks=extra)
    w = C.parse_wav(blob)
    assert (w.numchannels, w.samplerate, w.bitspersample, w.numsamples, w.blockalign) == (2, 8000, 16, 1001, 4)
    assert np.array_equal(C.pcm_from_wav(w), pcm)
    meta = C.pack_metadata(w.chunks)
    assert len(meta) == w.metadatasize
    assert C.unpack_metadata(meta) == w.chunks
    assert C.rebuild_wav(C.unpack_metadata(meta), w.data) == blob
    # trailing chunk after the data chunk, 8-bit mono (even byte count: the reference's seek past a
    # non-final data chunk is not word-aligned, wav.cpp:243-245, so odd sizes derail its own parser)
    pcm8 = synth_pcm(776, 1, 6, 8000, bits=8)
    b8 = C.wav_bytes_from_pcm(pcm8, 8000, 8)
    b8 = b8[:4] + (len(b8) - 8 + 12).to_bytes(4, "little") + b8[8:] + b"cue " + (4).to_bytes(4, "little") + b"\1\2\3\4"
    w8 = C.parse_wav(b8)
    assert w8.numsamples == 776 and np.array_equal(C.pcm_from_wav(w8), pcm8)
    assert C.rebuild_wav(C.unpack_metadata(C.pack_metadata(w8.chunks)), w8.data) == b8
    # odd number of 8-bit samples in a final data chunk: pad byte restored
    b9 = C.wav_bytes_from_pcm(synth_pcm(777, 1, 7, 8000, bits=8), 8000, 8)
    w9 = C.parse_wav(b9)
    assert w9.numsamples == 777 and C.rebuild_wav(C.unpack_metadata(C.pack_metadata(w9.chunks)), w9.data) == b9


@pytest.mark.ref
def test_sac_file_is_read_by_the_genuine_reference(orc, ref, tmp_path):
    """A .sac file written by sac_amd.container (records from the oracle) is opened by the genuine
    Sac::ReadSACHeader / ReadMD5 and decoded frame by frame by the genuine FrameCoder: header fields,
    MD5, metadata bytes and PCM all agree."""
    import hashlib
    from sac_amd import container as C
    from sac_amd.synth import synth_pcm
    rate, maxlen = 8000, 1
    pcm = synth_pcm(2 * rate + 345, 2, 9, rate)
    blob = C.wav_bytes_from_pcm(pcm, rate, 16, extra_chunks=[(0x5453494C, b"INFOICMT\x04\x00\x00\x00test")])
    w = C.parse_wav(blob)
    recs, pos = [], 0
    while pos < w.numsamples:                                  # frame loop of Codec::EncodeFile without adapt_block
        n = min(maxlen * rate, w.numsamples - pos)
        recs.append(orc.encode_frame(pcm[:, pos: pos + n], frame_cfg("normal"), maxlen * rate)["record"])
        pos += n
    path = str(tmp_path / "t.sac")
    C.write_sac(path, w, maxlen, recs)
    hdr, md5, meta, dec, nframes = ref.read_sac(path)
    assert hdr == dict(numchannels=2, samplerate=rate, bitspersample=16, numsamples=w.numsamples, max_framelen=maxlen,
                       metadatasize=w.metadatasize)
    assert md5 == hashlib.md5(w.data).digest()
    assert meta == C.pack_metadata(w.chunks)
    assert nframes == len(recs) == 3
    assert np.array_equal(dec, pcm)
    h2, m2, chunks2, recs2 = C.read_sac(path)
    assert recs2 == recs and chunks2 == w.chunks and m2 == md5


def test_cpp_container_header_matches_python(tmp_path):
    """sacenc's C++ container layer (sacfile.h: WAV walk, metadata, SAC2 header, own MD5) writes the
    same header + MD5 bytes as sac_amd.container (hashlib MD5) -- no device involved."""
    from sac_amd import container as C
    from sac_amd.synth import synth_pcm
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sac_amd", "sacenc")
    assert os.path.exists(exe), "sac_amd/sacenc not built (make -C sac_amd/csrc)"
    import hashlib
    for seed, nch, n, bits, extra in ((1, 2, 1001, 16, [(0x5453494C, b"INFOISFT\x05\x00\x00\x00abcde")]), (2, 1, 777, 8, []), (3, 2, 64, 16, [(0x6B6E756A, b"xyz")])):
        blob = C.wav_bytes_from_pcm(synth_pcm(n, nch, seed, 8000, bits=bits), 8000, bits, extra_chunks=extra)
        wav = tmp_path / f"t{seed}.wav"; out = tmp_path / f"t{seed}.hdr"
        wav.write_bytes(blob)
        subprocess.run([exe, "--header-only", "--framelen=7", str(wav), str(out)], check=True)
        w = C.parse_wav(blob)
        assert out.read_bytes() == C.sac_header(w, 7) + hashlib.md5(w.data).digest()


def test_sacenc_list_reads_a_sac_file(orc, tmp_path):
    """`sacenc --listfull` (host only; the reference's --list / --listfull, cmdline.cpp:295-323 + Codec::ScanFrames) on a
    .sac file written by sac_amd.container from oracle records: header fields, MD5, frame count and block sizes."""
    import hashlib
    from sac_amd import container as C
    from sac_amd.synth import synth_pcm
    exe = os.path.join(ROOT, "sac_amd", "sacenc")
    rate, maxlen = 8000, 1
    pcm = synth_pcm(2 * rate + 345, 2, 9, rate)
    w = C.parse_wav(C.wav_bytes_from_pcm(pcm, rate, 16))
    recs, pos = [], 0
    while pos < w.numsamples:
        n = min(maxlen * rate, w.numsamples - pos)
        recs.append(orc.encode_frame(pcm[:, pos: pos + n], frame_cfg("normal"), maxlen * rate)["record"])
        pos += n
    path = str(tmp_path / "t.sac")
    C.write_sac(path, w, maxlen, recs)
    out = subprocess.run([exe, "--listfull", path], capture_output=True, text=True, check=True).stdout
    assert f"{rate}Hz 16 Bit  2 channel(s)  {w.numsamples} samples" in out
    assert "Audio MD5: " + "".join("%x" % b for b in hashlib.md5(w.data).digest()) in out
    assert f"Frames   {len(recs)}" in out
    for i, r in enumerate(recs):
        assert f"Frame {i + 1}: {int.from_bytes(r[:4], 'little')} samples" in out
        assert f"Channel 0: {int.from_bytes(r[4 + 232: 8 + 232], 'little')} bytes" in out


@pytest.mark.parametrize("taps", [(2304, 1280, 768, 256), (2305, 1280, 768, 256), (2304, 1281, 767, 9), (4608, 2560, 1536, 512),
                                  (4609, 8, 8, 8), (264, 2561, 15, 2), (8, 8, 1537, 513), (7, 6, 5, 4), (9216 - 1, 33, 4, 2)])
def test_canonical_cascade_layout_boundaries_emulated_vs_oracle(emu, orc, taps):
    """The canonical-order cascade at the capacity boundaries of its three layouts (one, two and four rounds over the
    lanes), with stage lengths just below / at / above 8 and ragged everywhere: emulated kernel body == oracle, bit for
    bit (the oracle itself is pinned to the genuine reference on the round-2 traces)."""
    from sac_amd.synth import synth_pcm
    taps = tuple(min(t, c) for t, c in zip(taps, (8192, 4096, 2048, 1024)))
    n = 160
    raw = synth_pcm(n, 1, 77 + sum(taps) % 13, 8000)
    g = np.ascontiguousarray(orc.profile()[:, 2].copy(), np.float32)
    g[28], g[29], g[30], g[37] = taps
    smp, stats = center_frame(raw)
    plpc = np.zeros((1, n)); psum = np.zeros((1, n)); err = np.zeros((1, n), np.int32); pred = np.zeros((1, n), np.int32)
    rc = emu.emu_predict(1, n, _vp(np.ascontiguousarray(smp, np.int32)), _vp(np.ascontiguousarray(stats, np.int32)), _vp(g), 0, n, 0, 4,
                         _vp(plpc), _vp(psum), _vp(err), _vp(pred))
    assert rc == 0
    pd, ol, om, oe = orc.predict_trace(smp, stats, g, 0, n, 0)
    assert np.array_equal(plpc.view(np.uint64), ol.view(np.uint64))
    assert np.array_equal(psum.view(np.uint64), (ol + om).view(np.uint64))
    assert np.array_equal(err, oe)


# one stage-length tuple per search layout of the launcher (lms_class_for, kernels_pred.hip), each at a capacity edge of
# that layout: 15 slots (8,4,2,1); 22 slots (6,10,4,2) (13,5,3,1) (14,2,4,2) (9,5,5,3) -- the ones that use the factored
# step-size table; 30 slots (12,12,4,2) (18,6,4,2) (15,8,6,1); 512 lanes (16,8,4,2)
_SEARCH_LAYOUT_TAPS = [(2048, 1024, 512, 256), (1536, 2560, 1024, 512), (3328, 1280, 768, 256), (3584, 512, 1024, 512),
                       (2304, 1280, 1280, 768), (3072, 3072, 1024, 512), (4608, 1536, 1024, 512), (3840, 2048, 1536, 256),
                       (4609, 1537, 1025, 513), (1537, 1025, 513, 257), (3329, 7, 769, 3)]


@pytest.mark.parametrize("taps", _SEARCH_LAYOUT_TAPS)
def test_search_cascade_layouts_emulated_vs_oracle(emu, orc, taps):
    """The search-mode cascade (k = optk) in every layout the launcher can pick, at the layout's capacity and one past the
    previous one's: emulated kernel body vs oracle.  The OLS stage is bit-exact; the NLMS sums are free-order (and, in the
    22-slot layouts, use the factored step-size table), so the stage sum is held to the search tolerance of DESIGN.md and
    the residuals to the few samples where that moves a rounding."""
    from sac_amd.synth import synth_pcm
    n = 2600
    raw = synth_pcm(n, 1, 500 + sum(taps) % 17, 8000)
    g = np.ascontiguousarray(orc.profile()[:, 2].copy(), np.float32)
    g[28], g[29], g[30], g[37] = taps
    smp, stats = center_frame(raw)
    plpc = np.zeros((1, n)); psum = np.zeros((1, n)); err = np.zeros((1, n), np.int32); pred = np.zeros((1, n), np.int32)
    rc = emu.emu_predict(1, n, _vp(np.ascontiguousarray(smp, np.int32)), _vp(np.ascontiguousarray(stats, np.int32)), _vp(g), 0, n, 1, 4,
                         _vp(plpc), _vp(psum), _vp(err), _vp(pred))
    assert rc == 0
    pd, ol, om, oe = orc.predict_trace(smp, stats, g, 0, n, 1)
    want = ol + om
    assert np.array_equal(plpc.view(np.uint64), ol.view(np.uint64))
    assert np.max(np.abs(psum - want) / (np.abs(want) + 1.0)) < 1e-9
    assert int((err != oe).sum()) <= n // 200 and int(np.max(np.abs(err - oe))) <= 1


@pytest.mark.parametrize("nA,nM0,opt", [(5, 0, 0), (16, 1, 1), (17, 0, 0), (24, 0, 1), (25, 7, 0), (32, 1, 1), (32, 8, 0), (32, 15, 1), (32, 16, 0),
                                        (32, 17, 1), (32, 24, 0), (32, 31, 1), (32, 32, 0), (32, 32, 1)])
def test_register_resident_ols_kernel_body_vs_oracle(emu, orc, nA, nM0, opt):
    """The one-wave OLS kernel body (right-looking LDL^T on register rows, run-time column loop with rotating register
    slots; covariance rows in registers) at regressor lengths around every capacity class
    boundary, k = 1 and k = 4: p_lpc bit-identical to the oracle (itself pinned to the genuine reference)."""
    from sac_amd.synth import synth_pcm
    n = 220
    raw = synth_pcm(n, 1, 300 + nA + nM0, 8000)
    g = np.ascontiguousarray(orc.profile()[:, 2].copy(), np.float32)
    g[24], g[9] = nA, nM0
    g[28], g[29], g[30], g[37] = 64, 32, 16, 4          # short cascade: this test is about stage 1
    smp, stats = center_frame(raw)
    plpc = np.zeros((1, n)); psum = np.zeros((1, n)); err = np.zeros((1, n), np.int32); pred = np.zeros((1, n), np.int32)
    rc = emu.emu_predict(1, n, _vp(np.ascontiguousarray(smp, np.int32)), _vp(np.ascontiguousarray(stats, np.int32)), _vp(g), 0, n, opt, 4,
                         _vp(plpc), _vp(psum), _vp(err), _vp(pred))
    assert rc == 0
    pd, ol, om, oe = orc.predict_trace(smp, stats, g, 0, n, opt)
    assert np.array_equal(plpc.view(np.uint64), ol.view(np.uint64))
    assert np.array_equal(err, oe) or opt          # search evaluations: free-order cascade sums (tolerance elsewhere)


def _pack_items(orc, spec, n_of, seed0):
    """work-items for emu_ols_pack: spec = [(nch, slot, nA or nB.., extra)], returns the ctypes argument arrays and the oracle's p_lpc"""
    from sac_amd.synth import synth_pcm
    items = []
    for i, (nch, slot, n_self, n_other) in enumerate(spec):
        n = n_of(i)
        raw = synth_pcm(n, nch, seed0 + i, 8000)
        g = np.ascontiguousarray(orc.profile()[:, 2].copy(), np.float32)
        if slot == 0:
            g[24], g[9] = n_self, n_other                     # ch0: nA own samples + nM0 of the other channel
        else:
            g[25] = n_self; g[26] = min(n_other, 32); g[27] = max(n_other - 32, 0)    # ch1: nB + nS0 + nS1
        g[0] = orc.profile()[0, 0] + (orc.profile()[0, 1] - orc.profile()[0, 0]) * (0.2 + 0.05 * (i % 9))     # distinct regularisers / forgetting factors
        g[28], g[29], g[30], g[37] = 64, 32, 16, 4
        smp, stats = center_frame(raw)
        items.append((nch, slot, n, np.ascontiguousarray(smp, np.int32), np.ascontiguousarray(stats, np.int32), g))
    return items


@pytest.mark.parametrize("cls,opt", [(0, 1), (0, 0), (1, 1), (1, 0), (2, 1), (2, 0)])
def test_packed_ols_kernel_body_vs_oracle(emu, orc, cls, opt):
    """pred_ols_pack.h: four (16 lanes each) resp. two (32 lanes each) work-items per wave -- regressor lengths, frame lengths,
    channel slots and OLS coefficients differ inside a wave, one group left empty -- p_lpc of every item bit-identical to the
    oracle (itself pinned to the genuine reference), k = 1 (final pass) and k = 4 (search)."""
    nmax = (16, 24, 32)[cls]
    lens = [nmax, nmax - 1, 5, nmax - 7, 8, nmax - 3, 4, 9, nmax]           # last wave only partly filled
    spec = []
    for i, ln in enumerate(lens):
        if i % 3 == 2 and ln >= 6:
            spec.append((2, 1, ln // 2, ln - ln // 2))           # stereo, slot 1: own + other-channel samples
        elif i % 3 == 1 and ln >= 6:
            spec.append((2, 0, ln - 2, 2))                       # stereo, slot 0 with nM0 = 2
        else:
            spec.append((1, 0, ln, 0))
    items = _pack_items(orc, spec, lambda i: 150 + 17 * (i % 4), 700 + 10 * cls)
    cnt = len(items)
    IP = ctypes.POINTER(ctypes.c_int32); FP = ctypes.POINTER(ctypes.c_float); DP = ctypes.POINTER(ctypes.c_double)
    outs = [np.zeros(it[2]) for it in items]
    arr = lambda T, vals: (T * cnt)(*vals)
    rc = emu.emu_ols_pack(cls, cnt, arr(ctypes.c_int, [it[0] for it in items]), arr(ctypes.c_int, [it[3].shape[1] for it in items]),
                          arr(IP, [it[3].ctypes.data_as(IP) for it in items]), arr(IP, [it[4].ctypes.data_as(IP) for it in items]),
                          arr(FP, [it[5].ctypes.data_as(FP) for it in items]), arr(ctypes.c_int, [it[1] for it in items]),
                          arr(ctypes.c_int, [it[2] for it in items]), opt, 4, arr(DP, [o.ctypes.data_as(DP) for o in outs]))
    assert rc == 0
    for i, it in enumerate(items):
        nch, slot, n, smp, stats, g = it
        pd, ol, om, oe = orc.predict_trace(smp, stats, g, 0, n, opt)
        # file channel of predictor slot `slot`: ch_ref = 0 for these profiles (coefficient 27 >= 0)
        assert np.array_equal(outs[i].view(np.uint64), ol[slot if nch == 2 else 0].view(np.uint64)), (i, spec[i])


def test_decoder_body_emulated_inverts_the_golden_streams(emu, orc, golden):
    """The entropy-DEcoder kernel body (coder_stream_dec: RangeCoderSH::DecodeBitOne + BitplaneCoder::Decode [+ MapEncoder::
    Decode], vle.cpp:233-261, map.cpp:87-101) turns the genuine reference's bytes back into the s2u values / used flags."""
    fwd, inv = orc.domain_tables()
    u = np.ascontiguousarray(golden["coder/s2u"], np.int32)
    mb = int(golden["coder/maxbpn"][0])
    payload = np.ascontiguousarray(golden["coder/bytes"], np.uint8)
    out = np.zeros(u.size, np.int32)
    used = emu.emu_bitplane_decode(_vp(payload), payload.size, u.size, mb, None, _vp(fwd), _vp(inv), _vp(out))
    assert np.array_equal(out, u)
    assert used <= payload.size + 4          # the decoder reads up to four bytes past the flush (BufIO's zero fill)
    # mapped stream of the golden sparse frame: map header + remapped residual
    raw = golden["frame/sparse16_normal/raw"][0]
    smp, stats = center_frame(raw[None, :])
    err, pred = orc.predict_frame(smp, stats, golden["profile"][:, 2].copy(), 0, raw.size, 0)
    r, s2m, mbm, ul, uh = orc.remap(raw, pred[0], err[0])
    payload = np.ascontiguousarray(np.frombuffer(golden["frame/sparse16_normal/record"].tobytes()[4 + 232 + 18:], np.uint8))
    out = np.zeros(s2m.size, np.int32); flags = np.zeros(2 * 32769, np.uint8)
    emu.emu_bitplane_decode(_vp(payload), payload.size, s2m.size, mbm, _vp(flags), _vp(fwd), _vp(inv), _vp(out))
    assert np.array_equal(out, s2m)
    assert np.array_equal(flags[:32769], ul) and np.array_equal(flags[32769:], uh)
    # random residuals, ragged lengths (chunk boundaries, n < 64, single sample)
    rng = np.random.default_rng(3)
    for n in (1, 2, 63, 64, 65, 127, 129, 1000):
        e = np.rint(rng.laplace(size=n) * rng.choice([1, 20, 3000])).astype(np.int32)
        u = np.where(e < 0, -2 * e, np.where(e > 0, 2 * e - 1, 0)).astype(np.int32)
        mb = max(int(u.max()).bit_length() - 1, 0)
        payload = np.ascontiguousarray(np.frombuffer(orc.bitplane_encode(u, mb), np.uint8))
        out = np.zeros(n, np.int32)
        emu.emu_bitplane_decode(_vp(payload), payload.size, n, mb, None, _vp(fwd), _vp(inv), _vp(out))
        assert np.array_equal(out, u), n
