# round 4, call 8: packed OLS kernels after the register diet (chunked back-substitution, item constants in LDS, 2 waves / SIMD)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -p no:cacheprovider -k "predictor_stages or frame_records or evaluate_costs or random_profiles or edge_frames or kept_ols or 24bit_predictor" > $O/gputests_pack_subset2.log 2>&1; echo rc=$?; tail -3 $O/gputests_pack_subset2.log | cut -c1-200
for v in 1; do
  SACAMD_OLS_PACK=$v timeout 600 python tests/gpu_throughput.py 8192 12,16,20,24,28,32 > $O/throughput_pack${v}b.txt 2>&1
  echo == pack=$v; cat $O/throughput_pack${v}b.txt | cut -c1-200
done
