"""BASELINE configs[3] at the preset's FULL evaluation count (--best: E = 1000, CostBitplane objective over a 441 000-sample window, dds,8) on ONE
full-size frame, on the GPU box, in instalments (sacamd_search_frames_resume): the 125 lock-step generations are latency-bound (~15-20 s each
whatever the batch), more than one time-boxed call holds, so the search state travels in a file between calls.
    python tests/gpu_best_e1000.py <state file> <log json> [seconds this call may spend]
When the search is done the record is written (Predict without a search + Encode + WriteEncoded for the profile found) and compared with the
genuine reference's (tests/golden/ref_golden_r6.npz: best_s16_e1000, made from oracle/_ref with the reference's own 8 evaluation threads)."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sac_amd.api as api
from golden_cases import FULL_FRAMESIZE, config34_full_cases

state_path, log_path = sys.argv[1], sys.argv[2]
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 600.0
t_start = time.time()
g6 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_golden_r6.npz"))
name = "best_s16_e1000"
raw, cfg = config34_full_cases()[name]
assert hashlib.sha256(raw.astype(np.int16).tobytes()).digest() == g6[f"cfg/{name}/raw_sha256"].tobytes()
gcfg = api.Cfg(cfg.optimize, cfg.sparse_pcm, cfg.zero_mean, cfg.reset, cfg.fraction, cfg.maxnfunc, cfg.num_threads, cfg.sigma, cfg.optk, cfg.cost, 0)
ctx = api.Context(raw.shape[0], FULL_FRAMESIZE, 1)
ctx.upload_i32([raw], FULL_FRAMESIZE)
log = json.load(open(log_path)) if os.path.exists(log_path) else {"case": name, "calls": []}
state = open(state_path, "rb").read() if os.path.exists(state_path) else None
call = {"started_with_state_bytes": len(state) if state else 0, "instalments": []}
done, prof = False, None
while not done and time.time() - t_start < budget:
    t = time.time()
    prof, state, done = ctx.search_frames_resume(gcfg, 4, state)
    open(state_path, "wb").write(state)
    call["instalments"].append({"generations": 4, "seconds": round(time.time() - t, 2), "done": done})
    print(f"instalment: {time.time() - t:.1f} s, done {done}", flush=True)
call["seconds"] = round(time.time() - t_start, 1)
log["calls"].append(call)
if done:
    fin = api.Cfg(0, cfg.sparse_pcm, cfg.zero_mean, cfg.reset, cfg.fraction, cfg.maxnfunc, cfg.num_threads, cfg.sigma, cfg.optk, cfg.cost, 0)
    t = time.time()
    recs, _ = ctx.encode_frames(fin, profiles=prof)
    log["final_pass_and_coder_seconds"] = round(time.time() - t, 1)
    log["record_len"] = len(recs[0]); log["record_len_reference"] = int(g6[f"cfg/{name}/record_len"][0])
    log["record_sha256"] = hashlib.sha256(recs[0]).hexdigest(); log["record_sha256_reference"] = g6[f"cfg/{name}/record_sha256"].tobytes().hex()
    log["profile_equals_reference"] = bool(np.array_equal(prof[0], g6[f"cfg/{name}/profile"]))
    log["record_equals_reference"] = log["record_sha256"] == log["record_sha256_reference"]
    log["search_seconds_total"] = round(sum(c["seconds"] for c in log["calls"]), 1)
    log["reference_wall_seconds_8_threads_build_container"] = float(g6[f"cfg/{name}/wall_seconds_8_threads"][0])
    print("DONE", {k: log[k] for k in ("record_equals_reference", "profile_equals_reference", "search_seconds_total", "final_pass_and_coder_seconds")})
ctx.close()
json.dump(log, open(log_path, "w"), indent=1)
