#!/bin/bash
# One GPU-box session: tests, bench, kernel trace, PMC passes.  Run through gpurun from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a'
# Everything lands under gpurun_out/<tag>/.
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== quick parity check" | tee $OUT/summary.txt
timeout 400 python tools/gpu_check.py > $OUT/gpu_check.log 2>&1
RC=$?
grep -E "PARITY|MISMATCH|ALL OK|FAILURES|fault|batch 16" $OUT/gpu_check.log | cut -c1-200 | tee -a $OUT/summary.txt
if [ $RC -ne 0 ]; then echo "quick check failed (exit $RC): stopping here" | tee -a $OUT/summary.txt; tail -30 $OUT/gpu_check.log | cut -c1-300 | tee -a $OUT/summary.txt; exit 0; fi
echo "== pytest -m gpu" | tee -a $OUT/summary.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -x -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest.log | tee -a $OUT/summary.txt
echo "== pipeline_once B=64 / 256" | tee -a $OUT/summary.txt
timeout 300 python tools/pipeline_once.py 64 3 8 2>&1 | tail -3 | tee -a $OUT/summary.txt
timeout 300 python tools/pipeline_once.py 256 3 16 2>&1 | tail -3 | tee -a $OUT/summary.txt
echo "== bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?" | tee -a $OUT/summary.txt
tail -c 6000 $OUT/bench.json | tee -a $OUT/summary.txt
if [ "$2" != "noprof" ]; then
  echo "== kernel trace" | tee -a $OUT/summary.txt
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 256 2 16 > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
  DB=$(find $OUT/trace -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null && python tools/rocpd_timeline.py $DB > $OUT/fit_timeline.txt 2>&1
  head -30 $OUT/kernel_stats.md | tee -a $OUT/summary.txt
  cat $OUT/fit_timeline.txt | tee -a $OUT/summary.txt
  echo "== PMC passes (B=32)" | tee -a $OUT/summary.txt
  for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $G | tr ' ' '_' | cut -c1-40)
    (cd /tmp && timeout 300 rocprofv3 --pmc $G -d $GRAFT_REPO_ROOT/$OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 32 1 8 > $GRAFT_REPO_ROOT/$OUT/pmc_$N.log 2>&1)
    python tools/pmc_summary.py $OUT/pmc_$N >> $OUT/pmc_summary.md 2>&1
    echo >> $OUT/pmc_summary.md
  done
  cat $OUT/pmc_summary.md | tee -a $OUT/summary.txt
  # keep the merge-back small: drop the raw databases, keep the summaries
  find $OUT -name "*.db" -size +8M -delete
fi
echo "== done" | tee -a $OUT/summary.txt
