# round 4, call 18: back-substitution with loads two chunks ahead and the next row's first chunk ahead of the row's result store:
# final pass latency per class (few items: uncrowded), then 256 frames x 20 s (before: 121.5-123.8 s)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for m in "16/64" "24/56" "32/48" "16/40"; do timeout 300 python tests/gpu_finalpass.py 32 60000 "$m" 2>&1 | grep -v "^frames" | cut -c1-120; done | tee $O/finalpass_latency_bwd2.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 500 -p no:cacheprovider -k "predictor_stages or frame_records or random_profiles or gpu_decoder_inverts" 2>&1 | tail -2
SACAMD_TRACE=1 timeout 900 python bench.py --frames 256 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 --no-extras > $O/bench_256_bwd2.json 2> $O/bench_256_bwd2.err
python - <<PY
import json
d=json.loads(open("$O/bench_256_bwd2.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
grep "steps 882000" $O/bench_256_bwd2.err | tail -8 | cut -c1-150
