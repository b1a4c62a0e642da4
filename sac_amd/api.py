"""ctypes binding of libsac_amd.so (the C ABI in include/sac_amd.h) plus a thin, FrameCoder-shaped
Python host layer used by tests and bench.py.  There is no CPU fallback: if the HIP library is
missing, or no gfx950 device is present, construction fails loudly."""
from __future__ import annotations

import ctypes
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")   # must be set before the HIP runtime initialises
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_longlong, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SACAMD_LIB_PATH") or os.path.join(HERE, "libsac_amd.so")   # override: A/B builds of the library (tools/build_variant.sh)
NUM_COEFS = 58
COST_L1, COST_RMS, COST_ENTROPY, COST_GOLOMB, COST_BITPLANE = 0, 1, 2, 3, 4
SEARCH_DDS, SEARCH_DE, SEARCH_CMA = 0, 1, 2

ABI_SYMBOLS = [
    "sacamd_ctx_create", "sacamd_ctx_destroy", "sacamd_last_error", "sacamd_default_profile",
    "sacamd_default_cfg", "sacamd_frames_upload_i32", "sacamd_frames_upload_s16",
    "sacamd_frames_attach_s16_device", "sacamd_analyse", "sacamd_get_stats", "sacamd_evaluate",
    "sacamd_predict_final", "sacamd_get_residuals", "sacamd_encode", "sacamd_get_encoded",
    "sacamd_encode_frames", "sacamd_debug_predict", "sacamd_debug_bitplane", "sacamd_debug_cost",
    "sacamd_plan_subframes", "sacamd_subframes_from_states", "sacamd_kernel_times", "sacamd_class_times", "sacamd_eval_stats", "sacamd_debug_ols_profile", "sacamd_abi_version", "sacamd_progress", "sacamd_search_frames", "sacamd_assign_frames",
    "sacamd_decode_frames", "sacamd_comm_unique_id", "sacamd_comm_create", "sacamd_comm_destroy", "sacamd_comm_last_error",
    "sacamd_gather_records", "sacamd_gather_records_via", "sacamd_plan_cascade_streams", "sacamd_debug_libm", "sacamd_predictor_streams", "sacamd_get_encoded_variant", "sacamd_get_residuals_map",
    "sacamd_search_frames_resume", "sacamd_search_state_bytes",
]


class SacAmdError(RuntimeError):
    pass


class Cfg(ctypes.Structure):
    """sacamd_cfg == FrameCoder::tsac_cfg + toptim_cfg flattened (libsac/libsac.h:19-44)."""
    _fields_ = [("optimize", c_int), ("sparse_pcm", c_int), ("zero_mean", c_int), ("reset", c_int),
                ("fraction", c_double), ("maxnfunc", c_int), ("num_threads", c_int), ("sigma", c_double),
                ("optk", c_int), ("optimize_cost", c_int), ("optimize_search", c_int)]


_PRESETS = {  # cmdline.cpp:127-156
    "normal": (0, 0.0, 0, 0.2, COST_ENTROPY),
    "high": (1, 0.1, 100, 0.20, COST_ENTROPY),
    "veryhigh": (1, 0.2, 300, 0.25, COST_ENTROPY),
    "extrahigh": (1, 0.2, 600, 0.25, COST_ENTROPY),
    "best": (1, 0.5, 1000, 0.25, COST_BITPLANE),
    "insane": (1, 0.5, 1500, 0.25, COST_BITPLANE),
}


def make_cfg(mode="normal", num_threads=0, reset=1, sparse_pcm=1, zero_mean=1, fraction=None,
             maxnfunc=None, cost=None, optk=4, sigma=None, search=SEARCH_DDS) -> Cfg:
    """A preset of the reference's command line (cmdline.cpp:127-156) as a Cfg.  Note reset=1 (== --opt-reset) is the
    default HERE because the batch drivers run all frames in lock-step: with reset=0 (the reference's default, and what
    sacamd_default_cfg returns) a frame's search starts from the profile the caller passes in profiles_io, so chaining
    "best profile of frame f seeds frame f+1" is the caller's job, one frame of a file at a time."""
    o, f, e, s, c = _PRESETS[mode]
    return Cfg(o, sparse_pcm, zero_mean, reset, f if fraction is None else fraction,
               e if maxnfunc is None else maxnfunc, num_threads, s if sigma is None else sigma, optk,
               c if cost is None else cost, search)


_lib = None


ABI_VERSION = 7


class PredTParam(ctypes.Structure):
    """sacamd_pred_tparam (include/sac_amd.h) == Predictor::tparam (libsac/pred.h:11-27), flattened"""
    _fields_ = ([(k, c_int) for k in ("nA", "nB", "nM0", "nS0", "nS1", "k")] + [("vn0", c_int * 4), ("vn1", c_int * 4)] +
                [(k, c_double * 4) for k in ("vmu0", "vmu1", "vmudecay0", "vmudecay1", "vpowdecay0", "vpowdecay1")] +
                [(k, c_double) for k in ("lambda0", "lambda1", "ols_nu0", "ols_nu1", "mu_mix0", "mu_mix1", "mu_mix_beta0", "mu_mix_beta1",
                                         "beta_sum0", "beta_pow0", "beta_add0", "beta_sum1", "beta_pow1", "beta_add1")] +
                [("ch_ref", c_int), ("bias_mu0", c_double), ("bias_mu1", c_double), ("bias_scale0", c_int), ("bias_scale1", c_int), ("lm_n", c_int),
                 ("lm_alpha", c_double), ("proj_alpha0", c_double), ("proj_alpha1", c_double)])


def tparam_from_profile(coefs, optimize: bool, optk: int = 4) -> PredTParam:
    """FrameCoder::SetParam (libsac.cpp:37-92): 58 float32 coefficients -> Predictor::tparam"""
    g = np.asarray(coefs, np.float32)
    G = lambda i: float(g[i])                       # profile.Get(i): the float32 value, widened
    R = lambda i: int(np.sign(g[i]) * np.floor(abs(float(g[i])) + 0.5))      # std::round: half away from zero
    t = PredTParam()
    t.k = optk if optimize else 1
    t.vn0[:] = [R(28), R(29), R(30), R(37)]; t.vn1[:] = [R(31), R(32), R(33), R(38)]
    t.vmu0[:] = [G(2) / float(t.vn0[0]), G(3) / float(t.vn0[1]), G(4) / float(t.vn0[2]), G(5) / float(t.vn0[3])]
    t.vmu1[:] = [G(14) / float(t.vn1[0]), G(15) / float(t.vn1[1]), G(16) / float(t.vn1[2]), G(17) / float(t.vn1[3])]
    t.vmudecay0[:] = [G(6), G(39), G(46), G(47)]; t.vpowdecay0[:] = [G(7), G(8), G(50), G(51)]
    t.vmudecay1[:] = [G(18), G(40), G(48), G(49)]; t.vpowdecay1[:] = [G(19), G(20), G(21), G(52)]
    t.lambda0, t.ols_nu0, t.lambda1, t.ols_nu1 = G(0), G(1), G(12), G(13)
    t.mu_mix0, t.mu_mix_beta0, t.mu_mix1, t.mu_mix_beta1 = G(10), G(11), G(22), G(23)
    t.nA, t.nB, t.nS0, t.nS1, t.nM0 = R(24), R(25), R(26), R(27), R(9)
    t.beta_sum0, t.beta_pow0, t.beta_add0 = G(34), G(35), G(36)
    t.beta_sum1, t.beta_pow1, t.beta_add1 = G(53), G(54), G(55)
    t.proj_alpha0, t.proj_alpha1 = G(56), G(57)
    t.lm_n, t.lm_alpha = R(41), G(42)
    t.bias_mu0, t.bias_mu1 = G(43), G(44)
    t.bias_scale0 = t.bias_scale1 = R(45)
    t.ch_ref = 0
    if t.nS1 < 0:
        t.nS1 = -t.nS1; t.ch_ref = 1
    return t


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SacAmdError(f"{LIB_PATH} not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.sacamd_last_error.restype = c_char_p
        _lib.sacamd_last_error.argtypes = [c_void_p]
        _lib.sacamd_ctx_destroy.argtypes = [c_void_p]
        if _lib.sacamd_abi_version() != ABI_VERSION:
            v = _lib.sacamd_abi_version(); _lib = None
            raise SacAmdError(f"{LIB_PATH} has ABI version {v}, this binding needs {ABI_VERSION}: rebuild (make -C sac_amd/csrc)")
    return _lib


def _vp(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_void_p)


def default_profile() -> np.ndarray:
    """[58,3] = vmin, vmax, vdef (SacProfile::LoadBaseProfile)."""
    lib = load_library()
    out = np.zeros((3, NUM_COEFS), np.float32)
    lib.sacamd_default_profile(_vp(out[0]), _vp(out[1]), _vp(out[2]))
    return np.ascontiguousarray(out.T)


class SubFrame(ctypes.Structure):
    _fields_ = [("start", c_int), ("length", c_int), ("state", c_int)]


def subframes_from_states(block_state, block_len, min_frame_length):
    """Codec::PushState state machine (host only) -> [(start, length, state)]."""
    lib = load_library()
    st = np.ascontiguousarray(block_state, np.int32); bl = np.ascontiguousarray(block_len, np.int32)
    out = (SubFrame * max(1, len(st)))()
    cnt = c_int(0)
    rc = lib.sacamd_subframes_from_states(_vp(st), _vp(bl), len(st), int(min_frame_length), out, len(out), byref(cnt))
    if rc != 0:
        raise SacAmdError(f"sacamd_subframes_from_states failed ({rc})")
    return [(out[i].start, out[i].length, out[i].state) for i in range(cnt.value)]


def assign_frames(cost, world: int) -> np.ndarray:
    """owner[f] = rank of frame f: longest-first by cost onto the least loaded rank (host only, no device)."""
    lib = load_library()
    c = np.ascontiguousarray(cost, np.float64)
    owner = np.zeros(len(c), np.int32)
    rc = lib.sacamd_assign_frames(_vp(c), len(c), int(world), _vp(owner))
    if rc != 0:
        raise SacAmdError(f"sacamd_assign_frames failed ({rc})")
    return owner


def plan_cascade_streams(group, work, pools, small_work: float):
    """(order, stream) of the cascade launches of one predictor pass (sacamd_plan_cascade_streams; host only, no device):
    launch q of group[q] with work[q]; pools[g] = stream ids group g may use."""
    lib = load_library()
    g = np.ascontiguousarray(group, np.int32); w = np.ascontiguousarray(work, np.float64)
    begin = np.zeros(len(pools) + 1, np.int32)
    for i, p in enumerate(pools):
        begin[i + 1] = begin[i] + len(p)
    flat = np.ascontiguousarray([s for p in pools for s in p] or [0], np.int32)
    order = np.zeros(max(len(g), 1), np.int32); stream = np.zeros(max(len(g), 1), np.int32)
    lib.sacamd_plan_cascade_streams.argtypes = [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, ctypes.c_double, c_void_p, c_void_p]
    rc = lib.sacamd_plan_cascade_streams(len(g), _vp(g), _vp(w), len(pools), _vp(begin), _vp(flat), float(small_work), _vp(order), _vp(stream))
    if rc != 0:
        raise SacAmdError(f"sacamd_plan_cascade_streams failed ({rc})")
    return order[:len(g)].copy(), stream[:len(g)].copy()


# ---- multi-GPU record gather (include/sac_amd.h "multi-GPU record gather")
COMM_ID_BYTES = 128
_AG = ctypes.CFUNCTYPE(c_int, c_void_p, POINTER(c_longlong), POINTER(c_longlong), c_int)
_GB = ctypes.CFUNCTYPE(c_int, c_void_p)
_SR = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_void_p, c_longlong)


class TransportC(ctypes.Structure):
    """sacamd_transport"""
    _fields_ = [("self", c_void_p), ("rank", c_int), ("world", c_int), ("allgather_i64", _AG), ("group_begin", _GB),
                ("send", _SR), ("recv", _SR), ("group_end", _GB)]


def _pack_records(frame_ids, recs):
    ids = np.ascontiguousarray(frame_ids, np.int32)
    off = np.zeros(len(recs) + 1, np.int64)
    off[1:] = np.cumsum([len(r) for r in recs])
    blob = np.frombuffer(b"".join(recs), np.uint8).copy() if off[-1] else np.zeros(1, np.uint8)
    return ids, off, blob


def _default_cap(recs, total_frames):
    # rank 0's receive buffer when the caller gives no capacity: frames of one job are of similar size
    longest = max((len(r) for r in recs), default=0)
    return total_frames * (2 * longest + (1 << 16)) if longest else max(64, total_frames) * (4 << 20)


def _unpack_gathered(rc, out, out_off, total_frames, rank, err, as_bytes=True):
    if rc != 0:
        raise SacAmdError(f"record gather failed ({rc}): {err()}")
    if rank != 0:
        return None
    if not as_bytes:       # views into the receive buffer (no copy of a multi-gigabyte job's records)
        return [out[out_off[f]: out_off[f + 1]] for f in range(total_frames)]
    return [out[out_off[f]: out_off[f + 1]].tobytes() for f in range(total_frames)]


def comm_unique_id() -> bytes:
    """ncclGetUniqueId (rank 0); hand the bytes to every rank's Comm()."""
    lib = load_library()
    buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
    rc = lib.sacamd_comm_unique_id(buf)
    if rc != 0:
        raise SacAmdError(f"sacamd_comm_unique_id failed ({rc})")
    return bytes(buf)


class Comm:
    """RCCL communicator of the record gather: one per process / GPU (ncclCommInitRank)."""

    def __init__(self, device: int, rank: int, world: int, unique_id: bytes):
        self.lib = load_library()
        self.lib.sacamd_comm_last_error.restype = c_char_p
        self.lib.sacamd_comm_last_error.argtypes = [c_void_p]
        self.lib.sacamd_comm_destroy.argtypes = [c_void_p]
        self.rank, self.world = rank, world
        h = c_void_p()
        idb = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        rc = self.lib.sacamd_comm_create(int(device), int(rank), int(world), idb, byref(h))
        if rc != 0:
            raise SacAmdError(f"sacamd_comm_create failed ({rc})")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.sacamd_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_records(self, frame_ids, recs, total_frames: int, cap: int = None, as_bytes: bool = True):
        """Collective.  recs: this rank's frame records, frame_ids their global frame numbers.  Rank 0 returns the list of
        all total_frames records in frame order (what WriteEncoded appends to the file; bytes, or uint8 array views of the
        receive buffer with as_bytes=False), other ranks None."""
        ids, off, blob = _pack_records(frame_ids, recs)
        if self.rank == 0:
            cap = int(cap) if cap is not None else _default_cap(recs, total_frames)
            out = np.zeros(cap, np.uint8); out_off = np.zeros(total_frames + 1, np.int64)
        else:
            cap, out, out_off = 0, None, None
        rc = self.lib.sacamd_gather_records(self.h, len(recs), _vp(ids), _vp(blob), _vp(off), int(total_frames), _vp(out),
                                            c_longlong(cap), _vp(out_off))
        return _unpack_gathered(rc, out, out_off, total_frames, self.rank, lambda: self.lib.sacamd_comm_last_error(self.h).decode(), as_bytes)


def gather_records_via(transport: TransportC, frame_ids, recs, total_frames: int, cap: int = None):
    """The same gather over a caller-supplied transport (sacamd_gather_records_via)."""
    lib = load_library()
    ids, off, blob = _pack_records(frame_ids, recs)
    if transport.rank == 0:
        cap = int(cap) if cap is not None else _default_cap(recs, total_frames)
        out = np.zeros(cap, np.uint8); out_off = np.zeros(total_frames + 1, np.int64)
    else:
        cap, out, out_off = 0, None, None
    rc = lib.sacamd_gather_records_via(byref(transport), len(recs), _vp(ids), _vp(blob), _vp(off), int(total_frames), _vp(out),
                                       c_longlong(cap), _vp(out_off))
    return _unpack_gathered(rc, out, out_off, total_frames, transport.rank, lambda: "transport / argument error")


class Context:
    """One device context: a batch of up to max_frames frames of nch channels."""

    def __init__(self, nch: int, max_framesize: int, max_frames: int, device: int = 0):
        self.lib = load_library()
        self.nch, self.max_framesize, self.max_frames = nch, max_framesize, max_frames
        h = c_void_p()
        rc = self.lib.sacamd_ctx_create(device, nch, max_framesize, max_frames, byref(h))
        if rc != 0:
            raise SacAmdError(f"sacamd_ctx_create failed ({rc}): no usable gfx950 device / HIP error")
        self.h = h
        self.nframes = 0
        self.numsamples = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.sacamd_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SacAmdError(f"sacamd error {rc}: {self.lib.sacamd_last_error(self.h).decode()}")

    # ---- staging
    def upload_i32(self, frames, framesize: int):
        """frames: list of int32 arrays [nch, n_f] (raw, un-centred PCM)."""
        nf = len(frames)
        ns = np.array([f.shape[1] for f in frames], np.int32)
        stride = int(ns.max())
        buf = np.zeros((nf, self.nch, stride), np.int32)
        for i, f in enumerate(frames):
            buf[i, :, : f.shape[1]] = f
        self._chk(self.lib.sacamd_frames_upload_i32(self.h, nf, framesize, _vp(buf), c_longlong(self.nch * stride),
                                                    c_longlong(stride), _vp(ns)))
        self.nframes, self.numsamples = nf, ns

    def upload_s16(self, interleaved: np.ndarray, frame_offset, numsamples, framesize: int):
        """interleaved: int16 [total_sample_frames, nch]; frame f = rows frame_offset[f] .. +numsamples[f]."""
        il = np.ascontiguousarray(interleaved, np.int16)
        fo = np.ascontiguousarray(frame_offset, np.int64)
        ns = np.ascontiguousarray(numsamples, np.int32)
        self._chk(self.lib.sacamd_frames_upload_s16(self.h, len(ns), framesize, _vp(il), _vp(fo), _vp(ns)))
        self.nframes, self.numsamples = len(ns), ns

    def attach_s16_device(self, dev_ptr: int, frame_offset, numsamples, framesize: int):
        fo = np.ascontiguousarray(frame_offset, np.int64)
        ns = np.ascontiguousarray(numsamples, np.int32)
        self._chk(self.lib.sacamd_frames_attach_s16_device(self.h, len(ns), framesize, c_void_p(dev_ptr), _vp(fo), _vp(ns)))
        self.nframes, self.numsamples = len(ns), ns

    # ---- path stages
    def analyse(self, cfg: Cfg):
        self._chk(self.lib.sacamd_analyse(self.h, byref(cfg)))

    def stats(self) -> np.ndarray:
        out = np.zeros((self.nframes, self.nch, 4), np.int32)
        self._chk(self.lib.sacamd_get_stats(self.h, _vp(out)))
        return out

    def evaluate(self, cfg: Cfg, cand_frame, coefs) -> np.ndarray:
        cf = np.ascontiguousarray(cand_frame, np.int32)
        g = np.ascontiguousarray(coefs, np.float32).reshape(len(cf), NUM_COEFS)
        costs = np.zeros(len(cf))
        self._chk(self.lib.sacamd_evaluate(self.h, byref(cfg), len(cf), _vp(cf), _vp(g), _vp(costs)))
        return costs

    def predict_final(self, cfg: Cfg, coefs):
        g = np.ascontiguousarray(coefs, np.float32).reshape(self.nframes, NUM_COEFS)
        self._chk(self.lib.sacamd_predict_final(self.h, byref(cfg), _vp(g)))

    def residuals(self, frame: int):
        n = int(self.numsamples[frame])
        err = np.zeros((self.nch, n), np.int32); pred = np.zeros((self.nch, n), np.int32)
        s2u = np.zeros((self.nch, n), np.int32); mb = np.zeros(self.nch, np.int32)
        self._chk(self.lib.sacamd_get_residuals(self.h, frame, _vp(err), _vp(pred), _vp(s2u), _vp(mb)))
        return err, pred, s2u, mb

    def encode(self, cfg: Cfg):
        self._chk(self.lib.sacamd_encode(self.h, byref(cfg)))

    def encoded(self, frame: int, ch: int):
        n = int(self.numsamples[frame])
        out = np.zeros(n * 4 + 40000, np.uint8)
        ln, mp, mb = c_int(0), c_int(0), c_int(0)
        self._chk(self.lib.sacamd_get_encoded(self.h, frame, ch, _vp(out), out.size, byref(ln), byref(mp), byref(mb)))
        return out[: ln.value].tobytes(), mp.value, mb.value

    def encoded_variant(self, frame: int, ch: int, variant: int):
        """(bytes, maxbpn) of the Normal (0) / Mapped (1) coder stream of a channel, b"" if that variant was not coded (enc_temp1 / enc_temp2)"""
        n = int(self.numsamples[frame])
        out = np.zeros(n * 4 + 40000, np.uint8)
        ln, mb = c_int(0), c_int(0)
        self._chk(self.lib.sacamd_get_encoded_variant(self.h, frame, ch, variant, _vp(out), out.size, byref(ln), byref(mb)))
        return out[: ln.value].tobytes(), mb.value

    def residuals_map(self, frame: int):
        """(s2u_error_map [nch, n], maxbpn_map [nch]) after encode() with sparse_pcm (CalcRemapError)"""
        n = int(self.numsamples[frame])
        m = np.zeros((self.nch, n), np.int32); mb = np.zeros(self.nch, np.int32)
        self._chk(self.lib.sacamd_get_residuals_map(self.h, frame, _vp(m), _vp(mb)))
        return m, mb

    def encode_frames(self, cfg: Cfg, profiles=None):
        """Predict + Encode + WriteEncoded for every staged frame -> (list of records, profiles)."""
        prof = np.zeros((self.nframes, NUM_COEFS), np.float32)
        if profiles is None:
            prof[:] = default_profile()[:, 2]
        else:
            prof[:] = np.asarray(profiles, np.float32).reshape(self.nframes, NUM_COEFS)
        cap = int(self.numsamples.astype(np.int64).sum()) * self.nch * 4 + self.nframes * (4096 * 2 + 70000)
        out = np.zeros(cap, np.uint8)
        off = np.zeros(self.nframes + 1, np.int64)
        self._chk(self.lib.sacamd_encode_frames(self.h, byref(cfg), _vp(prof), _vp(out), c_longlong(cap), _vp(off)))
        recs = [out[off[f]: off[f + 1]].tobytes() for f in range(self.nframes)]
        return recs, prof

    def search_frames_resume(self, cfg: Cfg, max_generations: int, state=None, profiles=None):
        """The DDS search in instalments (sacamd_search_frames_resume): -> (best profiles so far, state blob, done)."""
        prof = np.zeros((self.nframes, NUM_COEFS), np.float32)
        prof[:] = default_profile()[:, 2] if profiles is None else np.asarray(profiles, np.float32).reshape(self.nframes, NUM_COEFS)
        self.lib.sacamd_search_state_bytes.restype = c_longlong
        cap = int(self.lib.sacamd_search_state_bytes(self.h))
        buf = np.zeros(cap, np.uint8)
        n = c_longlong(0)
        if state is not None:
            buf[: len(state)] = np.frombuffer(state, np.uint8); n = c_longlong(len(state))
        done = c_int(0)
        self._chk(self.lib.sacamd_search_frames_resume(self.h, byref(cfg), _vp(prof), int(max_generations), _vp(buf), c_longlong(cap), byref(n), byref(done)))
        return prof, buf[: n.value].tobytes(), bool(done.value)

    def decode_frames(self, recs, framesize):
        """Frame records (bytes, as encode_frames returns them / as they lie in a .sac file) -> (list of PCM arrays
        [nch, numsamples] int32, profiles [nframes, 58]): ReadEncoded + Decode + Unpredict of every frame on the GPU."""
        nf = len(recs)
        blob = np.frombuffer(b"".join(recs), np.uint8).copy()
        off = np.zeros(nf + 1, np.int64)
        off[1:] = np.cumsum([len(r) for r in recs])
        out = np.zeros((nf, self.nch, self.max_framesize), np.int32)
        ns = np.zeros(nf, np.int32)
        prof = np.zeros((nf, NUM_COEFS), np.float32)
        self._chk(self.lib.sacamd_decode_frames(self.h, nf, int(framesize), _vp(blob), _vp(off), _vp(out), c_longlong(self.nch * self.max_framesize),
                                                c_longlong(self.max_framesize), _vp(ns), _vp(prof)))
        return [out[f, :, : ns[f]].copy() for f in range(nf)], prof

    # ---- adaptive sub-frame split + batch file driver (Codec::Analyse / Codec::EncodeFile's frame loop)
    def plan_subframes(self, pcm, blocksamples, min_frame_length, samples_read=None):
        """pcm [nch, n] int32 (raw, un-centred) -> [(start, length, state)] as Codec::Analyse cuts one read."""
        pcm = np.ascontiguousarray(pcm, np.int32)
        nch, n = pcm.shape
        sr = n if samples_read is None else int(samples_read)
        out = (SubFrame * 64)()
        cnt = c_int(0)
        self._chk(self.lib.sacamd_plan_subframes(self.h, _vp(pcm), c_longlong(n), nch, sr, int(blocksamples), int(min_frame_length),
                                                 out, 64, byref(cnt)))
        return [(out[i].start, out[i].length, out[i].state) for i in range(cnt.value)]

    def plan_file(self, pcm, rate, max_framelen=20, adapt_block=True):
        """Frame list of one file as Codec::EncodeFile produces it (libsac.cpp:782-829): reads of
        max_framelen*rate samples, each cut into sub-frames (3 s blocks) -> [(start, length)] in file order."""
        pcm = np.asarray(pcm)
        total = pcm.shape[1]
        maxfs = int(max_framelen) * int(rate)
        frames = []
        pos = 0
        while pos < total:
            n = min(maxfs, total - pos)
            if adapt_block:
                subs = self.plan_subframes(pcm[:, pos: pos + n], 3 * rate, 3 * rate)
            else:
                subs = [(0, n, 0)]
            frames += [(pos + s, ln) for s, ln, _ in subs]
            pos += n
        return frames

    def encode_pcm_files(self, files, rate, cfg: Cfg, max_framelen=20, adapt_block=True):
        """Batch driver: every frame of every file in one staged batch (frames are independent with
        cfg.reset=1).  files: list of int [nch, n] arrays.  Returns per file the list of frame records."""
        if cfg.optimize and not cfg.reset:
            raise SacAmdError("encode_pcm_files runs all frames of all files as one lock-step batch: frames cannot inherit the "
                              "previous frame's profile (cfg.reset=0 == the reference without --opt-reset); pass reset=1, or chain "
                              "encode_frames(profiles=...) per file yourself")
        plans = [self.plan_file(f, rate, max_framelen, adapt_block) for f in files]
        frames, owner = [], []
        for fi, (f, plan) in enumerate(zip(files, plans)):
            for s, ln in plan:
                frames.append(np.ascontiguousarray(np.asarray(f)[:, s: s + ln], np.int32)); owner.append(fi)
        out = [[] for _ in files]
        for b0 in range(0, len(frames), self.max_frames):
            chunk = frames[b0: b0 + self.max_frames]
            self.upload_i32(chunk, int(max_framelen) * int(rate))
            recs, _ = self.encode_frames(cfg)
            for k, r in enumerate(recs):
                out[owner[b0 + k]].append(r)
        return out, plans

    # ---- parity taps
    def debug_predict(self, frame, coefs, start, n, optimize, optk=4):
        g = np.ascontiguousarray(coefs, np.float32)
        plpc = np.zeros((self.nch, n)); psum = np.zeros((self.nch, n))
        err = np.zeros((self.nch, n), np.int32); pred = np.zeros((self.nch, n), np.int32)
        self._chk(self.lib.sacamd_debug_predict(self.h, frame, _vp(g), start, n, int(optimize), optk, _vp(plpc), _vp(psum),
                                                _vp(err), _vp(pred)))
        return plpc, psum, err, pred

    def predictor_streams(self, src, range4, tp: "PredTParam"):
        """Predictor surface (libsac/pred.h:9-42): src [nch, n] = the slot-0 / slot-1 channel's mean-removed samples, range4 =
        (r0.lo, r0.hi, r1.lo, r1.hi); returns (pd, p_lpc, p_lms), each [nch, n] by predictor slot."""
        a = np.ascontiguousarray(src, np.int32)
        nch, n = a.shape
        r4 = np.ascontiguousarray(range4, np.int32)
        pd = np.zeros((nch, n)); pl = np.zeros((nch, n)); pm = np.zeros((nch, n))
        self._chk(self.lib.sacamd_predictor_streams(self.h, nch, _vp(a[0]), _vp(a[1]) if nch == 2 else None, n, _vp(r4), byref(tp),
                                                    _vp(pd), _vp(pl), _vp(pm)))
        return pd, pl, pm

    def debug_bitplane(self, s2u, maxbpn) -> bytes:
        u = np.ascontiguousarray(s2u, np.int32)
        out = np.zeros(u.size * 4 + 4096, np.uint8)
        ln = c_int(0)
        self._chk(self.lib.sacamd_debug_bitplane(self.h, _vp(u), u.size, maxbpn, _vp(out), out.size, byref(ln)))
        return out[: ln.value].tobytes()

    def debug_libm(self, kind, x, y=None) -> np.ndarray:
        """device exp (kind 0) / pow (1) / PredictLaplace(avg_sum, bpn) (2) on arrays of arguments"""
        xa = np.ascontiguousarray(x, np.float64)
        ya = np.ascontiguousarray(np.zeros_like(xa) if y is None else y, np.float64)
        out = np.zeros(xa.size)
        self._chk(self.lib.sacamd_debug_libm(self.h, int(kind), _vp(xa), _vp(ya), xa.size, _vp(out)))
        return out

    def debug_cost(self, kind, err) -> float:
        e = np.ascontiguousarray(err, np.int32)
        c = c_double(0)
        self._chk(self.lib.sacamd_debug_cost(self.h, kind, _vp(e), e.size, byref(c)))
        return c.value

    def ols_profile(self, on=True):
        out = np.zeros(16, np.uint64)
        self._chk(self.lib.sacamd_debug_ols_profile(self.h, int(on), _vp(out)))
        return out

    def eval_stats(self, reset=True):
        """(channel evaluations requested by the search, of those answered from the per-batch memo)."""
        out = np.zeros(2, np.int64)
        self._chk(self.lib.sacamd_eval_stats(self.h, _vp(out), int(reset)))
        return int(out[0]), int(out[1])

    def class_times(self, reset=True):
        """per kernel instance: {(kind, class): (ms, launches, item_steps, fp64 flops)}, kind 'ols' | 'lms'."""
        out = np.zeros(128)
        self._chk(self.lib.sacamd_class_times(self.h, _vp(out), out.size, int(reset)))
        o = out.reshape(32, 4)
        res = {("ols", c): tuple(o[c]) for c in range(16) if o[c, 1] > 0}
        res.update({("lms", c): tuple(o[16 + c]) for c in range(16) if o[16 + c, 1] > 0})
        return res

    def progress(self):
        """(phase, generation) of an encode_frames call running on this context in another thread."""
        ph, gen = c_int(0), c_int(0)
        self.lib.sacamd_progress(self.h, byref(ph), byref(gen))
        return ph.value, gen.value

    def kernel_times(self, reset=True):
        out = np.zeros(16)
        self._chk(self.lib.sacamd_kernel_times(self.h, _vp(out), int(reset)))
        names = ["analyse", "tables", "ols", "lms", "bias", "cost", "s2u_remap", "coder"]
        return {n: {"ms": out[i], "launches": int(out[8 + i])} for i, n in enumerate(names)}
