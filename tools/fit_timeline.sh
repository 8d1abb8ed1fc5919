export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02k
mkdir -p $OUT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 256 2 16 > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
cd $GRAFT_REPO_ROOT && python tools/rocpd_timeline.py $DB
find $OUT -name "*.db" -delete
