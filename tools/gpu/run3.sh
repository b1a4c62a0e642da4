set -x
cd $GRAFT_REPO_ROOT
timeout 1700 python bench.py --steps 4 --warmup 1 --pipeline 2 --no-cpu-baseline --budget-s 900 > gpurun_out/bench_r3b_pipe2.json 2> gpurun_out/bench_r3b_pipe2.err
tail -c 3000 gpurun_out/bench_r3b_pipe2.json
