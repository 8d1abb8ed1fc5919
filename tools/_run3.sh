cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 300 python tools/fq_prof.py 64 2>&1 | tee gpurun_out/r03c/fq_prof.txt
