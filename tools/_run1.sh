set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 500 python tools/gpu_check.py > gpurun_out/r03a/gpu_check.log 2>&1; echo "gpu_check exit $?"
grep -E "PARITY|MISMATCH|ALL OK|FAILURES" gpurun_out/r03a/gpu_check.log | cut -c1-160
bash tools/abn.sh "default noex" 4 2>&1 | tee gpurun_out/r03a/abn.txt
timeout 300 python tools/pipeline_once.py 256 3 16 2>&1 | tail -3 | tee gpurun_out/r03a/pipe.txt
