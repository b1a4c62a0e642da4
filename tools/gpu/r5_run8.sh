# round 5, call 8: grid kernel for 24 / 32 taps against the packed kernels; latency and throughput
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
( echo "== packed kernels"; timeout 100 python tests/gpu_ols_latency.py 24,32; timeout 100 python tests/gpu_throughput.py 2048,8192 24,32
  echo "== grid NB = 3 / 4"; SACAMD_OLS_GRID_SHORT=1 timeout 100 python tests/gpu_ols_latency.py 24,32; SACAMD_OLS_GRID_SHORT=1 timeout 100 python tests/gpu_throughput.py 2048,8192 24,32 ) > $O/ols_grid_short.txt 2>&1
cut -c1-118 $O/ols_grid_short.txt
