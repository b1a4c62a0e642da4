# round 5, call 5: grid OLS kernel v3 (backward solve: 16 stream values per register, v_fmac_f64_dpp row_newbcast): probe, latency, throughput, parity subset
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 30 ./tools/probe_newbcast > $O/probe_newbcast.txt 2>&1; cat $O/probe_newbcast.txt
timeout 120 python tests/gpu_ols_latency.py 40,48,56,64 > $O/ols_grid_latency_v3.txt 2>&1
timeout 200 python tests/gpu_throughput.py 2048,8192 40,48,56,64 > $O/ols_grid_throughput_v3.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu -x -k "random_profiles or predictor_stages or frame_records or warm_start or headline" > $O/gputests_03_grid_v3_subset.log 2>&1
cat $O/ols_grid_latency_v3.txt; grep -v "^$" $O/ols_grid_throughput_v3.txt | cut -c1-100; tail -5 $O/gputests_03_grid_v3_subset.log
