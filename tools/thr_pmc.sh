#!/bin/bash
# HBM traffic of the threshold pass (160 x 1080p frames per launch): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate
# passes over tools/thr_only.py, plus a kernel trace for the launch duration.  Prints the per-launch means.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-thrpmc}
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C -d $OUT/$C -o pmc -- python $GRAFT_REPO_ROOT/tools/thr_only.py > $OUT/$C.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$C k_threshold
  find $OUT/$C -name "*.db" -delete
done
tail -1 $OUT/FETCH_SIZE.log
