# round 5, call 12: stage lengths of the search's cascade items (128 frames) for the design of the 128-lane layouts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
SACAMD_TRACE=1 SACAMD_DUMP_VN=/tmp/vn.txt timeout 600 python bench.py --frames 128 --steps 1 --warmup 0 --budget-s 0 --no-cpu-baseline --verify-sample 0 --no-extras > /dev/null 2> /dev/null
python - <<'PY'
import numpy as np
v=np.loadtxt('/tmp/vn.txt',dtype=np.int64); np.save('gpurun_out/r05/search_vn_128.npy', v.astype(np.int16)); print(v.shape)
PY
