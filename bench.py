#!/usr/bin/env python3
"""bench.py -- SAC encode hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (analyse -> DDS search -> final prediction -> bitplane/SSE
range coding -> frame records on the host [-> RCCL gather to rank 0]) over one batch of synthetic
16-bit / 44.1 kHz stereo frames, `--high` preset (fraction 0.1, 100 evaluations, sigma 0.2,
entropy cost) with the search run as 8-candidate DDS generations and --opt-reset
(== reference `--high --opt-cfg=dds,8 --opt-reset`).  Inputs (interleaved int16 PCM) are resident
in HBM before the timed region starts.  Frames shard across ranks with no data-path collective
(weak scaling: every GPU gets --frames frames); the only collective is the final record gather.

Every step stages the PCM again (which clears the library's per-batch memo of channel evaluations), so
each step performs every distinct evaluation of its own search; nothing is carried from step to step.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); every context here
# uses 5 streams (main + one per kernel class) and several contexts may run concurrently.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RATE = 44100
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# algorithmic HBM bytes per predictor channel-step of each stage kernel (DESIGN.md "Data layout")
STAGE_BYTES = {"ols": 4 + 4 + 8,     # own + other-channel PCM (int32), p_lpc out (fp64)
               "lms": 8 + 4 + 8,     # p_lpc in, PCM target, p_lpc+p_lms out
               "bias": 8 + 4 + 4}    # p in, PCM, residual out


def make_batch(nframes, seconds, seed0):
    from sac_amd.synth import synth_pcm

    n = int(seconds * RATE)
    frames = [synth_pcm(n, 2, seed=seed0 + i, rate=RATE) for i in range(nframes)]
    il = np.concatenate([f.T.astype(np.int16) for f in frames], axis=0)   # [nframes*n, 2] interleaved
    return frames, np.ascontiguousarray(il), n


def shard_frames(total_frames, rank, world):
    """Frames of a corpus are independent units (--opt-reset): contiguous block per rank."""
    per = (total_frames + world - 1) // world
    lo = min(rank * per, total_frames)
    return list(range(lo, min(lo + per, total_frames)))


def gather_records(recs, rank, world, device):
    """Variable-length gather of frame records to rank 0 (RCCL on GPUs, gloo in the CPU tests):
    all_gather of the per-frame lengths, then one gather of the padded payloads."""
    import torch
    import torch.distributed as dist

    blob = b"".join(recs)
    meta = torch.tensor([len(recs), len(blob)], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)                     # ranks may hold different numbers of frames
    max_n = int(max(int(m[0].item()) for m in metas))
    mx = int(max(int(m[1].item()) for m in metas))
    lens = torch.zeros(max_n, dtype=torch.int64, device=device)
    lens[: len(recs)] = torch.tensor([len(r) for r in recs], dtype=torch.int64)
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    tots = [m[1:2] for m in metas]
    buf = torch.zeros(max(mx, 1), dtype=torch.uint8, device=device)
    if blob:
        buf[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    res = []
    for r in range(world):
        b = out[r][: int(tots[r].item())].cpu().numpy().tobytes()
        o = 0
        for ln in all_lens[r].tolist()[: int(metas[r][0].item())]:
            res.append(b[o: o + ln]); o += ln
    return res


def cpu_baseline(seconds=3.0, nthreads=8, nframes=3):
    """Reference `--high --opt-cfg=dds,N --opt-reset` CPU encode on a bounded sample: `nframes`
    stereo frames of `seconds` s with max frame length == `seconds` s, so the search window is the
    same 10 % of the frame and the evaluations-per-sample ratio equals the full-size workload's
    (about 15 s of CPU work on one core)."""
    from oracle_api import Checker, frame_cfg, ref_available
    from sac_amd.synth import synth_pcm

    kind = "reference" if ref_available() else "port"
    chk = Checker("ref" if kind == "reference" else "orc")
    n = int(seconds * RATE)
    cfg = frame_cfg("high", num_threads=nthreads, reset=1)
    dt, nbytes, nsamp = 0.0, 0, 0
    for i in range(nframes):
        raw = synth_pcm(n, 2, seed=4242 + i, rate=RATE)
        t = time.time()
        r = chk.encode_frame(raw, cfg, n)
        dt += time.time() - t
        nbytes += len(r["record"]); nsamp += raw.size
    return {"value": nsamp / dt / 1e6, "unit": "MSamples/s", "cores": 1, "kind": kind,
            "seconds": dt, "bps": 8 * nbytes / nsamp,
            "sample": f"{nframes} stereo frames of {seconds:g} s 44.1 kHz/16-bit, --high --opt-cfg=dds,{nthreads} --opt-reset, "
                      f"max frame length {seconds:g} s (search window 10 % of the frame as in the 20 s workload); "
                      + ("genuine reference objects (oracle/_ref), candidates evaluated serially on 1 core"
                         if kind == "reference" else "oracle restatement, 1 core")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--frames", type=int, default=int(os.environ.get("SAC_BENCH_FRAMES", 0)),
                    help="frames per GPU per step (0 = auto: 384, fewer when steps+warmup > 1 so that the run stays within ~15 min)")
    ap.add_argument("--seconds", type=float, default=float(os.environ.get("SAC_BENCH_SECONDS", 20.0)), help="frame length")
    ap.add_argument("--dds-n", type=int, default=8, help="DDS candidates per generation (--opt-cfg=dds,N)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("SAC_BENCH_GROUPS", 1)),
                    help="independent frame groups run concurrently on one GPU (own context + HIP streams each)")
    ap.add_argument("--mode", default="high")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true", help="decode every record with the oracle afterwards (slow)")
    args = ap.parse_args()
    if args.frames <= 0:
        # one step of F 20-s frames costs about 60 s of latency-bound final pass + coder plus ~0.24 s per frame
        nrun = max(1, args.steps + args.warmup)
        per_step = 900.0 / nrun
        args.frames = int(min(384, max(32, (per_step - 60.0 * args.seconds / 20.0) / (0.24 * args.seconds / 20.0))))

    import torch

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import sac_amd.api as api
    from concurrent.futures import ThreadPoolExecutor

    frames, il, n = make_batch(args.frames, args.seconds, seed0=1000 + 1000 * rank)
    d_pcm = torch.from_numpy(il).to(device)            # interleaved L/R int16, resident in HBM
    torch.cuda.synchronize()
    framesize = int(20 * RATE) if args.seconds >= 20 else n   # reference: max_framelen(20 s) * rate
    cfg = api.make_cfg(args.mode, num_threads=args.dds_n, reset=1)
    # The kernels are latency-bound recurrences (one wave or one workgroup per frame x candidate x
    # channel); independent groups of frames therefore run concurrently, each with its own context
    # (own device buffers and HIP streams), so that one group's low-occupancy phases (final pass,
    # range coder) overlap the others' search generations.
    ngroups = max(1, min(args.groups, args.frames))
    bounds = [round(g * args.frames / ngroups) for g in range(ngroups + 1)]
    groups = []
    for g in range(ngroups):
        lo, hi = bounds[g], bounds[g + 1]
        ctx = api.Context(2, max(n, 16), hi - lo, device=local_rank)
        groups.append((ctx, np.arange(lo, hi, dtype=np.int64) * n, np.full(hi - lo, n, np.int32)))
    pool = ThreadPoolExecutor(max_workers=ngroups)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_group(g):
        ctx, frame_off, nsamp = groups[g]
        ctx.attach_s16_device(d_pcm.data_ptr(), frame_off, nsamp, framesize)
        ctx.analyse(cfg)
        recs, prof = ctx.encode_frames(cfg)      # ctypes releases the GIL: groups overlap on the GPU
        return recs

    def step():
        recs = [r for part in pool.map(run_group, range(ngroups)) for r in part]
        if dist is not None:
            allrecs = gather_records(recs, rank, world, device)
        else:
            allrecs = recs
        return recs, allrecs

    def class_times():
        tot = {}
        for ctx, _, _ in groups:
            for k, v in ctx.class_times(reset=True).items():
                a = tot.get(k, (0.0, 0.0, 0.0))
                tot[k] = (a[0] + v[0], a[1] + v[1], a[2] + v[2])
        return tot

    def kernel_times():
        tot = None
        for ctx, _, _ in groups:
            kt = ctx.kernel_times(reset=True)
            if tot is None:
                tot = kt
            else:
                for k in kt:
                    tot[k]["ms"] += kt[k]["ms"]; tot[k]["launches"] += kt[k]["launches"]
        return tot

    for _ in range(args.warmup):
        step()
    kernel_times(); class_times()
    for ctx, _, _ in groups:
        ctx.eval_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        recs, allrecs = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kt = kernel_times()
    ct = class_times()
    ev = [ctx.eval_stats() for ctx, _, _ in groups]
    ev_req, ev_memo = sum(e[0] for e in ev), sum(e[1] for e in ev)

    samples_per_step = args.frames * n * 2 * world
    value = samples_per_step * args.steps / dt / 1e6
    if rank == 0:
        bps = 8 * sum(len(r) for r in allrecs) / samples_per_step
        # ---- roofline of the dominant kernel instance (HIP events on the stream each launch ran on)
        E, frac = cfg.maxnfunc, cfg.fraction
        nopt = min(n, int(np.ceil(framesize * frac)))
        ols_nmax = (16, 24, 32, 40, 48, 56, 64, 96)
        def kname(kind, cls):
            if kind == "ols":
                return f"k_ols<{64 if cls < 3 else 256},{ols_nmax[cls]}>"
            return f"k_lms<{cls}>"
        cands = {}
        for (kind, cls), (ms, launches, isteps) in ct.items():
            cands[kname(kind, cls)] = (ms, launches, STAGE_BYTES[kind] * isteps)
        cands["k_coder"] = (kt["coder"]["ms"], max(kt["coder"]["launches"], 1), (4 + bps / 8) * n * 2 * args.frames * args.steps)
        dom = max(cands, key=lambda k: cands[k][0])
        dms, dlaunch, dbytes = cands[dom]
        avg_s = dms / 1e3 / max(dlaunch, 1)
        achieved = dbytes / max(dlaunch, 1) / avg_s / 1e9 if avg_s > 0 else 0.0
        out = {
            "metric": "encode MSamples/s + bps, 16-bit/44.1kHz stereo, --high; 1/2/4/8 MI355X",
            "value": value, "unit": "MSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.frames} frames/GPU x {args.seconds:g} s stereo 16-bit 44.1 kHz, --{args.mode} "
                                   f"--opt-cfg=dds,{args.dds_n} --opt-reset, GPU bitplane coder (BASELINE configs[2])",
                       "frames_per_gpu": args.frames, "frame_seconds": args.seconds, "dds_n": args.dds_n,
                       "concurrent_groups": ngroups,
                       "parallelism": f"frames sharded over {world} GPU(s), RCCL record gather"},
            "bps": bps, "x_realtime": (args.frames * world * args.seconds * args.steps) / dt,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "launches": int(dlaunch), "avg_launch_ms": avg_s * 1e3, "algorithmic_bytes_per_launch": dbytes / max(dlaunch, 1),
                         "note": "latency/fp64-VALU-bound recurrences; HBM fraction is expected to be << 1 % (SURVEY.md 8d)"},
            # the search never computes an identical channel evaluation twice within one batch (memo cleared
            # by every staging, i.e. every step): requested vs answered without recomputation, this rank
            "search_channel_evaluations": {"requested": ev_req, "shared_or_memoised": ev_memo},
            "kernel_ms": {k: round(v["ms"], 2) for k, v in kt.items()},
            "kernel_launches": {k: v["launches"] for k, v in kt.items()},
            "kernel_instances_ms": {k: round(v[0], 2) for k, v in sorted(cands.items())},
            "kernel_instances_launches": {k: int(v[1]) for k, v in sorted(cands.items())},
            "kernel_instances_algorithmic_MB": {k: round(v[2] / 1e6, 1) for k, v in sorted(cands.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(nthreads=args.dds_n)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = value / cb["value"]
        if args.verify:
            from oracle_api import Checker
            orc = Checker("orc")
            ok = all(np.array_equal(orc.decode_frame(r, 2, max(n, 16))[0], f) for r, f in zip(recs, frames))
            out["verified_lossless"] = bool(ok)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
