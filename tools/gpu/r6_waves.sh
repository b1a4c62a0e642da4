# per-wave section cycle counters of the search cascade (A/B libraries built with -DSACAMD_EXP_PROF_WAVE=w): tools/gpu/r6_waves.sh <tag> <variants...>
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
tag=$1; shift
{ echo "== wave 0"; timeout 300 python tests/gpu_latency.py 2>&1 | grep -A1 "lms cycles" | grep -B1 "k=4" | grep -v "^--" | head -4
for v in "$@"; do echo "== wave ${v#w}"; SACAMD_LIB_PATH=$GRAFT_REPO_ROOT/sac_amd/libsac_amd_$v.so timeout 300 python tests/gpu_latency.py 2>&1 | grep -B1 "k=4" | grep "lms cycles" | head -2; done; } | tee $O/latency_sections_per_wave_$tag.txt
