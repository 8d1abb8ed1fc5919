#!/bin/bash
# One GPU-box session: tests, bench, kernel trace, PMC passes, latency, per-phase counters.  Run through gpurun from the repo root:
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r03_v1'
# Everything lands under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== quick parity check" | tee $OUT/summary.txt
timeout 400 python tools/gpu_check.py > $OUT/gpu_check.log 2>&1
RC=$?
grep -E "PARITY|MISMATCH|ALL OK|FAILURES|fault" $OUT/gpu_check.log | cut -c1-200 | tee -a $OUT/summary.txt
if [ $RC -ne 0 ]; then echo "quick check failed (exit $RC): stopping here" | tee -a $OUT/summary.txt; tail -30 $OUT/gpu_check.log | cut -c1-300 | tee -a $OUT/summary.txt; exit 0; fi
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q --maxfail=8 -x -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest.log | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -c 7000 $OUT/bench.json | tee -a $OUT/summary.txt
if [ "$2" != "noprof" ]; then
  echo "== kernel trace (256 distinct frames)" | tee -a $OUT/summary.txt
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 256 2 256 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
  DB=$(find $OUT/trace -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null && python tools/rocpd_timeline.py $DB > $OUT/fit_timeline.txt 2>&1
  head -32 $OUT/kernel_stats.md | tee -a $OUT/summary.txt
  cat $OUT/fit_timeline.txt | tee -a $OUT/summary.txt
  find $OUT/trace -name "*.db" -delete
  echo "== PMC passes (B=32)" | tee -a $OUT/summary.txt
  bash tools/pmc_round.sh $TAG/pmc "k_fit k_quad k_cc_ k_points k_scatter k_cluster_select k_decode k_reconcile k_prologue k_threshold" > /dev/null 2>&1
  cat $OUT/pmc/pmc_summary.md | cut -c1-260 | tee -a $OUT/summary.txt
  echo "== threshold PMC (160 frames per launch)" | tee -a $OUT/summary.txt
  bash tools/thr_pmc.sh $TAG/thrpmc 2>&1 | tee -a $OUT/summary.txt
  echo "== single-frame latency" | tee -a $OUT/summary.txt
  timeout 600 python tools/latency.py 2>&1 | grep "^{" | tee $OUT/latency.txt | cut -c1-260 | tee -a $OUT/summary.txt
  echo "== other configs" | tee -a $OUT/summary.txt
  timeout 600 python tools/config_rates.py 2>&1 | tail -8 | tee $OUT/config_rates.txt | tee -a $OUT/summary.txt
  echo "== per-phase counters (tools build)" | tee -a $OUT/summary.txt
  [ -f isaac_ros_apriltag_amd/libapriltag_amd_prof.so ] && timeout 300 python tools/fq_prof.py 64 2>&1 | grep -v amdgpu.ids | tee $OUT/fq_prof.txt | cut -c1-300 | tee -a $OUT/summary.txt
fi
echo "== done" | tee -a $OUT/summary.txt
