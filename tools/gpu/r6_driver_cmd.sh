# the driver's command on the round's final tree (2 of 20 steps fit bench.py's own wall budget)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_1536.json 2> $O/bench_driver_cmd_1536.err
echo "wall seconds: $(( $(date +%s) - t0 ))" | tee -a $O/bench_driver_cmd_1536.err
tail -c 600 $O/bench_driver_cmd_1536.json
