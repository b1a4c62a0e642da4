set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "predictor_stages or canonical or random_profiles or frame_records or kept_ols or search_memo or full_size_frames" 2>&1 | tail -5
timeout 300 python tests/gpu_latency.py 2>&1 | grep -v "lms cycles" | grep -v "8192\|3383" | head -40
timeout 600 python tests/gpu_throughput.py 8192 16,24,32,40,48,56,64 2>&1 | tail -7
