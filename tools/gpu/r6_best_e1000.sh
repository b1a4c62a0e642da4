# one call of the full-size --best E = 1000 search in instalments; the state blob travels in profiles/r06/best_e1000/ (gpurun_out/ is not sent to the box)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
[ -f profiles/r06/best_e1000/state.bin ] && cp profiles/r06/best_e1000/state.bin $O/best_e1000_state.bin
[ -f profiles/r06/best_e1000/log.json ] && cp profiles/r06/best_e1000/log.json $O/best_e1000_log.json
timeout 1000 python tests/gpu_best_e1000.py $O/best_e1000_state.bin $O/best_e1000_log.json ${1:-780} 2>&1 | grep -v "mse:" | tail -8
