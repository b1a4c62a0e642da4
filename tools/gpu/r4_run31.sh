# round 4, call 31: three cascade groups in the search (OLS classes 0-2 | 3-4 | 5-7); 768 frames, one step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 500 -p no:cacheprovider -k "frame_records or evaluate_costs or random_profiles or search_memo or headline_config" 2>&1 | tail -2
SACAMD_TRACE=1 timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 --no-extras > $O/bench_768_g3.json 2> $O/bench_768_g3.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_768_g3.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
grep "steps 882000" $O/bench_768_g3.err | tail -7 | cut -c1-140
