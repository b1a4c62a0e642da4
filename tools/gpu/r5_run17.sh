# round 5, call 17: BASELINE configs[3] with the preset's FULL evaluation count (--best: E = 1000, CostBitplane over a 441 000-sample window), 32 frames
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 2600 python tests/gpu_baseline_configs.py --frames-best 32 --full best_s16_e1000 > $O/configs3_full.json 2> $O/configs3_full.err
cut -c1-900 $O/configs3_full.json; grep -v mse $O/configs3_full.err | tail -3
