cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
timeout 500 python tools/gpu_check.py > gpurun_out/r03h/gpu_check.log 2>&1; echo "gpu_check exit $?"
grep -E "MISMATCH|ALL OK|FAILURES" gpurun_out/r03h/gpu_check.log | cut -c1-160
bash tools/abn.sh "default" 4 2>&1 | tee gpurun_out/r03h/abn.txt
export TMPDIR=/tmp
OUT=gpurun_out/r03h
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 256 2 16 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
DB=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null && python tools/rocpd_timeline.py $DB > $OUT/fit_timeline.txt 2>&1
head -16 $OUT/kernel_stats.md
cat $OUT/fit_timeline.txt
find $OUT -name "*.db" -size +8M -delete
