# round 3, second GPU call: new tests, final-pass grouping A/B with launch timeline, BASELINE configs[3]/[4] at real size, batch-size probe
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "library_is or de_and_cma or decode_cli or rccl or frame_records or costs_and_coder or baseline_configs or full_size_frames or wav_to_sac or sacenc_cli or batched_frames or gpu_decoder_inverts" > $O/gputests_new.log 2>&1; tail -5 $O/gputests_new.log
SACAMD_TRACE=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_groups_on.json 2> $O/bench_groups_on.trace
SACAMD_FINAL_GROUPS=0 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 > $O/bench_groups_off.json 2> $O/bench_groups_off.err
python - <<'PY'
import json
for f in ('on','off'):
    d=json.loads([l for l in open(f'gpurun_out/r03/bench_groups_{f}.json') if l.startswith('{')][-1])
    print(f, d['value'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d['kernel_ms'])
PY
grep "882000" $O/bench_groups_on.trace | tail -40
timeout 1500 python tests/gpu_baseline_configs.py > $O/configs34.json 2> $O/configs34.err; cat $O/configs34.json; tail -3 $O/configs34.err
timeout 900 python bench.py --frames 768 --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 > $O/bench_768x20s.json 2> $O/bench_768x20s.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_768x20s.json') if l.startswith('{')][-1])
print('768', d['value'], d['ms_per_step'], d['bps'], d['kernel_ms'])
PY
