# round 4, call 13: chained tail (per-group OLS -> cascade -> bias -> S2U -> remap -> coder) -- record parity subset, then 256 frames x 20 s
# stepwise vs chained with the launch timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x --timeout 900 -p no:cacheprovider -k "frame_records or batched_frames or full_size_frames or edge_frames or many_coder or baseline_configs_3 or wav_to_sac or sacenc_cli or 24bit_material or de_and_cma or batch_file_driver or genuine_reference" > $O/gputests_chain_subset.log 2>&1; echo rc=$?; tail -4 $O/gputests_chain_subset.log | cut -c1-200
for v in 1 0; do
  SACAMD_TAIL_STEPWISE=$v SACAMD_TRACE=1 timeout 900 python bench.py --frames 256 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_256_stepwise$v.json 2> $O/bench_256_stepwise$v.err
  echo == stepwise=$v; python - <<PY
import json
d=json.loads(open("$O/bench_256_stepwise$v.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
  grep "steps 882000\|lms class 1[0-3]" $O/bench_256_stepwise$v.err | tail -16 | cut -c1-150
done
