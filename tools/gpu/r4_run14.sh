# round 4, call 14: chained tail without host-blocking copies in the enqueue path: 256 frames x 20 s, timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in 0; do
  SACAMD_TAIL_STEPWISE=$v SACAMD_TRACE=1 timeout 900 python bench.py --frames 256 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_256_chain2.json 2> $O/bench_256_chain2.err
  python - <<PY
import json
d=json.loads(open("$O/bench_256_chain2.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
  grep "steps 882000\|lms class 1[0-3]" $O/bench_256_chain2.err | tail -18 | cut -c1-150
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 500 -p no:cacheprovider -k "frame_records or edge_frames or 24bit_material or batch_file_driver" 2>&1 | tail -2
