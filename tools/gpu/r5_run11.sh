# round 5, call 11: search-cascade work by one-wave slot need (trace histogram), 256 frames
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
SACAMD_TRACE=1 timeout 600 python bench.py --frames 256 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_256_hist.json 2> $O/bench_256_hist.err
grep "slot need" $O/bench_256_hist.err
python - $O/bench_256_hist.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(round(d['value'],3), 'MSamples/s', round(d['ms_per_step']/1e3,1), 's/step'); print(d['kernel_ms'])
PY
