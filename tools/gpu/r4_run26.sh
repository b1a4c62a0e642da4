# round 4, call 27: the same with the panel launches of the final pass submitted 5 ms ahead of the one-wave classes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -p no:cacheprovider -k "predictor_stages or canonical_cascade or frame_records or evaluate_costs or random_profiles or edge_frames or kept_ols or gpu_decoder_inverts or gpu_decoder_roundtrip or box_maximum or search_memo" 2>&1 | tail -2
SACAMD_TRACE=1 timeout 1200 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 --no-extras > $O/bench_768_stagger.json 2> $O/bench_768_stagger.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_768_stagger.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
grep "steps 882000" $O/bench_768_stagger.err | tail -8 | cut -c1-140
