#!/bin/bash
# round 6 GPU calls, one parameterised script: tools/gpu/r6_run.sh <step> (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
TP='1280,256,32,4;1500,2500,900,400;3300,1200,700,250;3000,2900,900,400'     # 15-slot, 22-slot (6,10,4,2), mostly 30-slot, 30-slot (12,12,4,2) [until the layouts were re-chosen the last one was 3900,1900,900,400]
case "$1" in
  sweep1)   # pipelined search sweep: parity subset, section cycles, saturated throughput (base + prefetch depth 4)
    timeout 900 python -m pytest tests -q -m gpu -x -k "random_profiles or predictor_stages or evaluate_costs or headline or kept_ols" > $O/gputests_01_sweep_subset.log 2>&1
    tail -3 $O/gputests_01_sweep_subset.log
    timeout 300 python tests/gpu_latency.py > $O/latency_sections_sweep1.txt 2>&1; grep -B1 "k=4" $O/latency_sections_sweep1.txt | cut -c1-150
    timeout 600 python tests/gpu_throughput.py 4096 "" "$TP" > $O/throughput_lms_sweep1.txt 2>&1; grep "16 taps" $O/throughput_lms_sweep1.txt | cut -c1-160
    SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_ahead4.so timeout 600 python tests/gpu_throughput.py 4096 "" "$TP" > $O/throughput_lms_sweep1_ahead4.txt 2>&1; grep "16 taps" $O/throughput_lms_sweep1_ahead4.txt | cut -c1-160
    ;;
  sweep2)   # the four stages as one pipeline; Predictor's three-argument constructor; one 768-frame step
    timeout 600 python -m pytest tests -q -m gpu -x -k "predictor_class or predictor_surface or predictor_stages or evaluate_costs" > $O/gputests_02_sweep2_subset.log 2>&1
    tail -3 $O/gputests_02_sweep2_subset.log
    timeout 300 python tests/gpu_latency.py > $O/latency_sections_sweep2.txt 2>&1; grep -B1 "k=4" $O/latency_sections_sweep2.txt | cut -c1-150 | head -12
    timeout 600 python tests/gpu_throughput.py 4096 "" "$TP" > $O/throughput_lms_sweep2.txt 2>&1; grep "16 taps" $O/throughput_lms_sweep2.txt | cut -c1-160
    timeout 1500 python bench.py --frames 768 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --verify-sample 0 > $O/bench_768_sweep2.json 2> $O/bench_768_sweep2.err
    tail -c 2500 $O/bench_768_sweep2.json
    ;;
  call3)    # loop-invariant parameters in LDS; new full-preset / instalment / run_single tests; one 768-frame step
    timeout 300 python tests/gpu_latency.py > $O/latency_sections_call3.txt 2>&1; grep -B1 "k=4" $O/latency_sections_call3.txt | cut -c1-150 | head -4; grep -A1 "3383" $O/latency_sections_call3.txt | cut -c1-150
    timeout 600 python tests/gpu_throughput.py 4096 "" "$TP" > $O/throughput_lms_call3.txt 2>&1; grep "16 taps" $O/throughput_lms_call3.txt | cut -c1-160
    timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 -k "instalments or run_single or full_preset or predictor_class or evaluate_costs or both_coder_variants or frame_records" > $O/gputests_03_call3_subset.log 2>&1
    tail -14 $O/gputests_03_call3_subset.log
    timeout 1500 python bench.py --frames 768 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --verify-sample 0 > $O/bench_768_call3.json 2> $O/bench_768_call3.err
    python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/bench_768_call3.json")); print(d["value"], d["ms_per_step"], d["bps"], d["kernel_ms"])
PY
    ;;
  call4)    # lane-map canonical cascade as a software pipeline (no spills): parity, per-layout latency, final pass in isolation, 1536-frame step
    timeout 900 python -m pytest tests -q -m gpu -x -k "canonical or predictor_stages or frame_records or decoder_inverts or full_size_records" > $O/gputests_04_canon.log 2>&1; tail -3 $O/gputests_04_canon.log
    for t in 1280,256,32,4 2500,1200,600,128 3383,1168,614,273 5400,64,16,8 6900,64,8,8; do timeout 200 python tests/gpu_dbg_canon.py $t 6000; done > $O/canon_latency_call4.txt 2>&1; cat $O/canon_latency_call4.txt
    timeout 600 python tests/gpu_finalpass.py 256 60000 "16/32;32/48;24/64" > $O/finalpass_call4.txt 2>&1; cat $O/finalpass_call4.txt | cut -c1-170
    timeout 1500 python bench.py --frames 1536 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --verify-sample 0 > $O/bench_1536_call4.json 2> $O/bench_1536_call4.err
    python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/bench_1536_call4.json")); print(d["value"], d["ms_per_step"], d["bps"], d["kernel_ms"]); print({k:round(v/1e3,1) for k,v in d["kernel_instances_ms"].items()})
PY
    ;;
  call5)    # head: batched LDS loads, readlane hand-over -- section cycles, throughput, canonical latency, parity subset
    timeout 300 python tests/gpu_latency.py > $O/latency_sections_$2.txt 2>&1; grep -B1 "k=4" $O/latency_sections_$2.txt | cut -c1-150 | head -4; grep -A1 "3383" $O/latency_sections_$2.txt | cut -c1-150 | head -4
    timeout 600 python tests/gpu_throughput.py 4096 "" "$TP" > $O/throughput_lms_$2.txt 2>&1; grep "16 taps" $O/throughput_lms_$2.txt | cut -c1-160
    for t in 1280,256,32,4 3383,1168,614,273 5400,64,16,8; do timeout 200 python tests/gpu_dbg_canon.py $t 6000; done > $O/canon_latency_$2.txt 2>&1; cat $O/canon_latency_$2.txt
    timeout 900 python -m pytest tests -q -m gpu -x -k "canonical or predictor_stages or frame_records or decoder_inverts or evaluate_costs or random_profiles or warm_start" > $O/gputests_05_$2.log 2>&1; tail -3 $O/gputests_05_$2.log
    ;;
  trace1536)   # one 1536-frame step with the launch trace on (per-class launch start / duration of every generation)
    SACAMD_TRACE=1 timeout 1500 python bench.py --frames 1536 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --verify-sample 0 > $O/bench_1536_$2.json 2> $O/bench_1536_$2.err
    grep "sacamd trace" $O/bench_1536_$2.err > $O/launch_trace_1536_$2.txt; wc -l $O/launch_trace_1536_$2.txt
    python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/r06/bench_1536_$2.json") if l.startswith("{")][-1]; print(d["value"], d["ms_per_step"], d["bps"], d["kernel_ms"])
PY
    ;;
  bench1536)
    timeout 1500 python bench.py --frames 1536 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --verify-sample 0 > $O/bench_1536_$2.json 2> $O/bench_1536_$2.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r06/bench_1536_$2.json")); print(d["value"], d["ms_per_step"], d["bps"], d["kernel_ms"]); print({k:round(v/1e3,1) for k,v in d["kernel_instances_ms"].items()})
PY
    ;;
  mtfac)    # factored step-size table + 22-slot layouts at three workgroups per CU: throughput, sections, every test that runs a search
    timeout 300 python tests/gpu_latency.py > $O/latency_sections_$2.txt 2>&1; grep -B1 "k=4" $O/latency_sections_$2.txt | cut -c1-150 | head -4
    timeout 600 python tests/gpu_throughput.py 4096 "" "$TP" > $O/throughput_lms_$2.txt 2>&1; grep "16 taps" $O/throughput_lms_$2.txt | cut -c1-160
    timeout 1800 python -m pytest tests -q -m gpu -x --durations=5 -k "frame_records or headline or evaluate_costs or random_profiles or warm_start or de_and_cma or wide or instalments or configs_3_and_4 or kept_ols or chain or smoke or framecoder_wrapper_writes or batch" > $O/gputests_06_$2.log 2>&1; tail -9 $O/gputests_06_$2.log
    ;;
  baselines)   # configs[3] / [4] and the default run_single search with the CPU reference timed on THIS box in the same run
    timeout 1500 python bench.py --mode veryhigh --frames 64 --steps 1 --warmup 0 --no-extras --no-all-cores --cpu-threads-only --verify-sample 2 > $O/baseline_veryhigh_64.json 2> $O/baseline_veryhigh_64.err
    timeout 1500 python bench.py --mode best --maxnfunc 100 --frames 64 --steps 1 --warmup 0 --no-extras --no-all-cores --cpu-threads-only --verify-sample 2 > $O/baseline_best_e100_64.json 2> $O/baseline_best_e100_64.err
    timeout 1500 python bench.py --dds-n 0 --frames 256 --steps 1 --warmup 0 --no-extras --no-all-cores --verify-sample 2 > $O/baseline_run_single_256.json 2> $O/baseline_run_single_256.err
    python - <<'PY'
import json
for n in ("veryhigh_64","best_e100_64","run_single_256"):
    try:
        d=json.load(open(f"gpurun_out/r06/baseline_{n}.json")); cb=d.get("cpu_baseline") or {}
        print(n, "GPU", round(d["value"],4), "MSamples/s", round(d["ms_per_step"]/1e3,1), "s bps", round(d["bps"],4), "| CPU", cb.get("value"), "cores", cb.get("cores"), "kind", cb.get("kind"), "ratio", round(d["value"]/cb["value"],1) if cb.get("value") else None, "rec0 equal", cb.get("same_record_as_gpu"))
    except Exception as e: print(n, "failed", e)
PY
    ;;
  bench768)
    timeout 1500 python bench.py --frames 768 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --verify-sample 0 > $O/bench_768_$2.json 2> $O/bench_768_$2.err
    tail -c 1500 $O/bench_768_$2.json
    ;;
esac
