# round 5, call 3: the 2D-cyclic one-wave OLS kernel (pred_ols_grid.h) against the round-4 kernels: latency, saturated throughput, parity subset
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
( echo "== round-4 kernels (SACAMD_OLS_GRID=0)"; SACAMD_OLS_GRID=0 timeout 120 python tests/gpu_ols_latency.py 40,48,56,64
  echo "== grid, 2 waves/SIMD"; timeout 120 python tests/gpu_ols_latency.py 40,48,56,64
  echo "== grid, 1 wave/SIMD"; SACAMD_LIB_PATH=$PWD/sac_amd/libsac_amd_g1.so timeout 120 python tests/gpu_ols_latency.py 40,48,56,64 ) > $O/ols_grid_latency.txt 2>&1
( echo "== round-4 kernels (SACAMD_OLS_GRID=0)"; SACAMD_OLS_GRID=0 timeout 200 python tests/gpu_throughput.py 512,2048,8192 40,48,56,64
  echo "== grid, 2 waves/SIMD"; timeout 200 python tests/gpu_throughput.py 512,2048,8192 40,48,56,64
  echo "== grid, 1 wave/SIMD"; SACAMD_LIB_PATH=$PWD/sac_amd/libsac_amd_g1.so timeout 200 python tests/gpu_throughput.py 512,2048,8192 40,48,56,64 ) > $O/ols_grid_throughput.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu -x -k "random_profiles or predictor_stages or frame_records or warm_start or headline" > $O/gputests_02_grid_subset.log 2>&1
cat $O/ols_grid_latency.txt; grep -v "^$" $O/ols_grid_throughput.txt | cut -c1-110; tail -5 $O/gputests_02_grid_subset.log
