# round 5, call 20: rocprofv3 --kernel-trace --stats of one step of the bench default; then the two PMC passes (FETCH_SIZE, WRITE_SIZE) on 64 x 20 s
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_1536x20s_profiled.json 2> $O/bench_1536x20s_profiled.err
for f in $(find /tmp/prof_full -name "*kernel_stats.csv"); do cp $f $O/kernel_stats_1536x20s.csv; done
for f in $(find /tmp/prof_full -name "*domain_stats.csv"); do cp $f $O/domain_stats_1536x20s.csv; done
head -12 $O/kernel_stats_1536x20s.csv | cut -c1-70,150-240
for c in FETCH_SIZE WRITE_SIZE; do
  SAC_BENCH_SYNTH_PROCS=1 timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python bench.py --frames 64 --seconds 20 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
done
python tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_hbm_64x20s.txt 2>&1
python tools/pmc_to_json.py $O/bench_pmc_FETCH_SIZE.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $O/pmc_hbm.json | tail -1
head -12 $O/pmc_hbm_64x20s.txt | cut -c1-100
