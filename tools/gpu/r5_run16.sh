# round 5, call 16: BASELINE configs[4] with the preset's FULL evaluation count (--veryhigh, E = 300) against the reference's records;
# sacenc (C++ host, no GPU_MAX_HW_QUEUES in its environment) against bench.py on the same 256 frames; search cascade group boundary A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python tests/gpu_baseline_configs.py --frames-vh 64 --full vh_m8_e300,vh_s16_e300 > $O/configs4_full.json 2> $O/configs4_full.err
cut -c1-700 $O/configs4_full.json
timeout 900 python tests/gpu_sacenc_vs_bench.py --frames 256 > $O/sacenc_vs_bench.json 2> $O/sacenc_vs_bench.err; cut -c1-900 $O/sacenc_vs_bench.json; tail -2 $O/sacenc_vs_bench.err
B="python bench.py --frames 768 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras"
SACAMD_FAST_OLS=5 timeout 600 $B > $O/bench_768_fast5.json 2> /dev/null
python - $O/bench_768_fast5.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], round(d['value'],3), 'MSamples/s', round(d['ms_per_step']/1e3,1), 's/step'); print(d['kernel_ms'])
PY
