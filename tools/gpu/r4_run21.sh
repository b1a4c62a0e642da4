# round 4, call 21: the three configs[3]/[4] full-size tests after the test's own hash fix, then bench.py smoke with the extras
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --timeout 1200 -p no:cacheprovider --durations=5 -k "baseline_configs_3_and_4_full_size" > $O/gputests_configs34.log 2>&1; echo rc=$?; tail -12 $O/gputests_configs34.log | cut -c1-200
timeout 600 python bench.py --frames 96 --seconds 4 --steps 2 --warmup 1 --no-all-cores --budget-s 500 > $O/bench_smoke.json 2> $O/bench_smoke.err; echo rc=$?
python - <<PY
import json
L=open("$O/bench_smoke.json").read().strip().split("\n")
d=json.loads(L[-1])
print(len(L), "lines;", {k: d.get(k) for k in ("value","steps","ms_per_step","small_batch","single_frame_s","speedup_vs_cpu_baseline","speedup_vs_reference_threads","verified_lossless")}, d["h2d"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
tail -3 $O/bench_smoke.err | cut -c1-300
