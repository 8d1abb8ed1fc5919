#!/bin/bash
# PMC passes over the B=32 pipeline (one counter group per pass; FETCH_SIZE and WRITE_SIZE cannot share one).  Run on the GPU
# box from the repo root: bash tools/pmc_round.sh <tag> ["kernel substrings"].  Writes gpurun_out/<tag>/pmc_summary.md
TAG=${1:-pmc}
PATS=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $G -d $OUT/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 32 1 8 > $OUT/pmc_$N.log 2>&1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_$N $PATS >> $OUT/pmc_summary.md 2>&1
  echo >> $OUT/pmc_summary.md
done
# FETCH / WRITE totals per kernel as JSON (bench.py's stage_roofline reads the committed copy under profiles/);
# tools/pipeline_once.py 32 1 8 runs 1 + 1 submissions of 32 frames
mkdir -p $OUT/pmc_rw && cp -r $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_rw/ 2>/dev/null
python $GRAFT_REPO_ROOT/tools/pmc_json.py $OUT/pmc_rw 32 $OUT/pipeline_pmc.json
find $OUT -name "*.db" -delete
cat $OUT/pmc_summary.md
