# round 5, call 4: grid OLS kernel v2 (grouped updates, branch-free stores, weights of the backward solve from LDS): latency + throughput, both wave builds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
( echo "== grid v2, 1 wave/SIMD build"; SACAMD_GRID_WAVES=1 timeout 120 python tests/gpu_ols_latency.py 40,48,56,64
  echo "== grid v2, 2 waves/SIMD build"; SACAMD_GRID_WAVES=2 timeout 120 python tests/gpu_ols_latency.py 40,48,56,64 ) > $O/ols_grid_latency_v2.txt 2>&1
( echo "== grid v2, 1 wave/SIMD build"; SACAMD_GRID_WAVES=1 timeout 200 python tests/gpu_throughput.py 2048,8192 40,48,56,64
  echo "== grid v2, 2 waves/SIMD build"; SACAMD_GRID_WAVES=2 timeout 200 python tests/gpu_throughput.py 2048,8192 40,48,56,64 ) > $O/ols_grid_throughput_v2.txt 2>&1
cat $O/ols_grid_latency_v2.txt; grep -v "^$" $O/ols_grid_throughput_v2.txt | cut -c1-100
