# round 4, call 29: default panel budget (13/20 of the slots) + 300 us head start for whole-CU cascade layouts; 768 frames, one step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 500 -p no:cacheprovider -k "predictor_stages or canonical_cascade or frame_records or evaluate_costs or random_profiles or edge_frames" 2>&1 | tail -2
SACAMD_TRACE=1 timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 --no-extras > $O/bench_768_head.json 2> $O/bench_768_head.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_768_head.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
print({k:v for k,v in d["kernel_instances_ms"].items() if "lms" in k})
PY
grep "steps 882000\|lms class 1[0-3].*items [0-9][0-9][0-9]" $O/bench_768_head.err | tail -12 | cut -c1-140
