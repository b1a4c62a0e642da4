# rocprofv3 --kernel-trace --stats of one step of the bench default; then the two PMC passes (FETCH_SIZE, WRITE_SIZE) on 64 x 20 s (separate runs, as the guide prescribes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_1536x20s_profiled.json 2> $O/bench_1536x20s_profiled.err
for f in $(find /tmp/prof_full -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_1536x20s.csv; done
for f in $(find /tmp/prof_full -name "*domain_stats.csv"); do cp $f $O/domain_stats_1536x20s.csv; done
head -14 $O/kernel_stats_1536x20s.csv | cut -c1-70,150-240
for c in FETCH_SIZE WRITE_SIZE; do
  SAC_BENCH_SYNTH_PROCS=1 timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py --frames 64 --seconds 20 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_hbm_64x20s.txt 2>&1
python tools/pmc_to_json.py $O/bench_pmc_FETCH_SIZE.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/pmc_hbm.json | tail -1
head -14 $O/pmc_hbm_64x20s.txt | cut -c1-110
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r06/bench_1536x20s_profiled.json") if l.startswith("{")][-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["launches"], r["avg_launch_ms"], r["achieved"], r["frac"])
PY
