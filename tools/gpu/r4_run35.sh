# round 4, call 35: 1536 frames per step on the final tree (one step, no warm-up step: first-launch costs included)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1000 python bench.py --frames 1536 --steps 1 --warmup 0 --no-cpu-baseline --verify-sample 0 --no-extras > $O/bench_1536_final_tree.json 2> $O/bench_1536_final_tree.err
tail -c 1500 $O/bench_1536_final_tree.json
