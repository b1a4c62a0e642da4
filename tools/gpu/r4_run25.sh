# round 4, call 25: one-wave OLS kernels with the covariance rows in registers up to 48 / 64 taps (default: up to 32, LDS triangle above)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for v in base m48 m64; do
  if [ $v = base ]; then unset SACAMD_LIB_PATH; else export SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_$v.so; fi
  SACAMD_OLS_PANEL_SLOTS=0 timeout 900 python tests/gpu_throughput.py 8192 40,48,56,64 > $O/throughput_mreg_$v.txt 2>&1
  echo == $v; cat $O/throughput_mreg_$v.txt | cut -c1-100
done
SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_m64.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 500 -p no:cacheprovider -k "predictor_stages or frame_records or random_profiles or evaluate_costs or kept_ols" 2>&1 | tail -2
