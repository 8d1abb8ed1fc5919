cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
export TMPDIR=/tmp
AMDAT_LIB= timeout 300 python tools/gpu_check.py 2>&1 | grep -E "MISMATCH|ALL OK|FAILURES" | cut -c1-160
for V in "" $1; do
  OUT=gpurun_out/r03i/t_${V:-def}
  (cd /tmp && AMDAT_LIB=$V timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT -o trace -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 256 2 16 > $GRAFT_REPO_ROOT/$OUT.log 2>&1)
  DB=$(find $OUT -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB $OUT.md > /dev/null
  echo "== ${V:-def}"; grep "prefilter" $OUT.md; python tools/rocpd_timeline.py $DB | head -1
  rm -rf $OUT
done
bash tools/abn.sh "default $1" 3 2>&1 | tee gpurun_out/r03i/abn.txt
