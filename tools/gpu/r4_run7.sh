# round 4, call 7: packed OLS kernels (2-4 work-items per wave) -- parity subset, then saturated throughput A/B and the final pass in isolation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -p no:cacheprovider -k "predictor_stages or canonical_cascade or frame_records or evaluate_costs or batched_frames or random_profiles or edge_frames or search_memo or kept_ols or 24bit_predictor or baseline_configs_3" > $O/gputests_pack_subset.log 2>&1; echo rc=$?; tail -5 $O/gputests_pack_subset.log | cut -c1-200
for v in 0 1; do
  SACAMD_OLS_PACK=$v timeout 600 python tests/gpu_throughput.py 8192 16,24,32 > $O/throughput_pack$v.txt 2>&1
  echo == pack=$v; cat $O/throughput_pack$v.txt | cut -c1-200
done
MIX="16/32;24/48;32/56;16/40;24/56;32/48;16/56;32/40;24/32;16/48;32/64;24/40"
for v in 0 1; do
  SACAMD_OLS_PACK=$v timeout 900 python tests/gpu_finalpass.py 768 40000 "$MIX" > $O/finalpass_pack$v.txt 2>&1
  echo == pack=$v; cat $O/finalpass_pack$v.txt | cut -c1-200
done
