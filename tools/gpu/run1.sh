set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in 1280,256,32,4 3000,500,100,50 3267,237,480,156 3383,1168,614,273 4728,1092,530,4 5000,1000,500,100; do
  timeout 300 python tests/gpu_dbg_canon.py $t 20000 2>&1 | tail -2
done
SACAMD_CANON_SYSTOLIC=1 timeout 300 python tests/gpu_dbg_canon.py 1280,256,32,4 20000 2>&1 | tail -1
SACAMD_CANON_SYSTOLIC=1 timeout 300 python tests/gpu_dbg_canon.py 3267,237,480,156 20000 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "canonical or predictor_stages or library or framecoder" 2>&1 | tail -5
