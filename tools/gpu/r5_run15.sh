# round 5, call 15: LDS-aware layout choice on the GPU + run_single measurement (256 frames)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -x -k "layout_choice or canonical_cascade or random_profiles or box_maximum" > $O/gputests_05_layout.log 2>&1; tail -3 $O/gputests_05_layout.log
timeout 1200 python tests/gpu_measure_r5.py --frames-dec 0 --frames-single 256 > $O/measure_r5_single.json 2> $O/measure_r5_single.err
cut -c1-1200 $O/measure_r5_single.json; grep -v mse $O/measure_r5_single.err | tail -3
