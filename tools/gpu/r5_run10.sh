# round 5, call 10: Predictor surface tests + parity subset on the current tree; then the bench default (1536 frames), one step, launch trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "predictor or random_profiles or frame_records or warm_start or library_is or framecoder" > $O/gputests_04_predictor.log 2>&1
tail -4 $O/gputests_04_predictor.log
SACAMD_TRACE=1 timeout 900 python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_1536_grid.json 2> $O/bench_1536_grid.err
python - $O/bench_1536_grid.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],3), 'MSamples/s', round(d['ms_per_step']/1e3,1), 's/step bps', round(d['bps'],4)); print(d['kernel_ms']); print(d['kernel_instances_ms'])
PY
grep "sacamd trace" $O/bench_1536_grid.err | grep "steps 882000\|lms class 1[0-3]" | tail -22
