# round 4, call 24: batch size: one step at 1152 and at 1536 frames x 20 s per GPU (default 768)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for f in 1152 1536; do
  timeout 1500 python bench.py --frames $f --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 --no-extras --budget-s 0 > $O/bench_${f}x20s.json 2> $O/bench_${f}x20s.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_${f}x20s.json") if l.startswith("{")][-1])
print($f, d["value"], d["ms_per_step"], d["bps"], d.get("verified_lossless"), d["kernel_ms"])
PY
done
