# round 3, fifth GPU call: final-pass OLS kernel choice at 768 frames (one-wave register-resident vs four-wave panel), same box
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sacenc_cli" > $O/gputests_sacenc.log 2>&1; tail -3 $O/gputests_sacenc.log
SACAMD_OLS_FINAL_PANEL=0 timeout 1200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 2 > $O/bench_768_final_onewave.json 2> $O/bench_768_final_onewave.err
timeout 1200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --verify-sample 0 > $O/bench_768_final_panel.json 2> $O/bench_768_final_panel.err
python - <<'PY'
import json
for f in ('onewave','panel'):
    d=json.loads([l for l in open(f'gpurun_out/r03/bench_768_final_{f}.json') if l.startswith('{')][-1])
    print(f, d['value'], d['ms_per_step'], d['bps'], d.get('verified_lossless'), d['kernel_ms'])
    print({k:v for k,v in d['kernel_instances_ms'].items() if 'ols' in k})
PY
# SQ issue counters per kernel (what the waves spend their cycles on), 64 x 20 s, one pass
export TMPDIR=/tmp
SAC_BENCH_SYNTH_PROCS=1 timeout 1200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_sq -o pmc -- python bench.py --frames 64 --seconds 20 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 > $O/bench_pmc_sq.json 2> $O/bench_pmc_sq.err
tail -2 $O/bench_pmc_sq.err
python tools/pmc_summary.py /tmp/pmc_sq > $O/pmc_sq_64x20s.txt 2>&1; head -40 $O/pmc_sq_64x20s.txt | cut -c1-220
