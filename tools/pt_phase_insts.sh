#!/bin/bash
# VALU instructions of k_points / k_cc_local up to each phase boundary (tools-only truncation variants pt1..pt3 built with
# -DAMDAT_PT_STOP=n -DAMDAT_CC_STOP=n).  Run on the GPU box from the repo root.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ptphase}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for V in pt1 pt2 pt3 ""; do
  N=${V:-full}
  AMDAT_LIB=$V timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 32 1 8 > $OUT/p_$N.log 2>&1
  echo "### after phase: $N" >> $OUT/pt_phase_insts.md
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/p_$N k_points k_cc_local >> $OUT/pt_phase_insts.md 2>&1
  find $OUT/p_$N -name "*.db" -delete
done
cat $OUT/pt_phase_insts.md
