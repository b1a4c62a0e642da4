# round 5, call 21: the corrected buffers test; BASELINE configs[3] (--best) at a tenth of the preset's evaluation count (E = 100), 64 frames
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests -q -m gpu -x -k "both_coder_variants or framecoder_wrapper" > $O/gputests_07_buffers.log 2>&1; tail -3 $O/gputests_07_buffers.log
timeout 1100 python tests/gpu_baseline_configs.py --frames-best 64 --full best_s16_e100 > $O/configs3_e100.json 2> $O/configs3_e100.err
cut -c1-1000 $O/configs3_e100.json; grep -v mse $O/configs3_e100.err | tail -2
