# round 3, third GPU call: 24-bit tests, step pipelining with a separate tail stream set
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "24bit or library_is or costs_and_coder or subframe_plan or frame_records or decode_cli" > $O/gputests_24bit.log 2>&1; tail -5 $O/gputests_24bit.log
SACAMD_TAIL_STREAMS=1 timeout 1500 python bench.py --pipeline 2 --steps 4 --warmup 1 --no-cpu-baseline --verify-sample 2 --budget-s 0 > $O/bench_pipeline2_tailstreams.json 2> $O/bench_pipeline2_tailstreams.err
python - <<'PY'
import json
for l in open('gpurun_out/r03/bench_pipeline2_tailstreams.json'):
    if l.startswith('{'):
        d=json.loads(l); print('pipe2', d['steps'], d['value'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d.get('kernel_ms'))
PY
tail -3 $O/bench_pipeline2_tailstreams.err
