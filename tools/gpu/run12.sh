# round 3, fourth GPU call: decoder grouping tests, cascade stream spreading A/B (384 frames, one step each)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decode_cli or decoder_groups or 24bit_subframe or gpu_decoder" > $O/gputests_dec.log 2>&1; tail -5 $O/gputests_dec.log
SACAMD_TRACE=1 timeout 900 python bench.py --frames 384 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_spread_on.json 2> $O/bench_spread_on.trace
SACAMD_LMS_STREAMS=0 timeout 900 python bench.py --frames 384 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 > $O/bench_spread_off.json 2> $O/bench_spread_off.err
python - <<'PY'
import json
for f in ('on','off'):
    d=json.loads([l for l in open(f'gpurun_out/r03/bench_spread_{f}.json') if l.startswith('{')][-1])
    print(f, d['value'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d['kernel_ms'])
PY
