# round 4, call 19: per-section cycle counters of the predictor kernels, single work-item (tests/gpu_latency.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python tests/gpu_latency.py > $O/latency_sections.txt 2>&1; cat $O/latency_sections.txt | cut -c1-170
