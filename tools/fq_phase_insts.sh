#!/bin/bash
# VALU instructions of the quad-fit kernel up to each phase boundary, per size class (B=32 sigma-2 frames): the variants
# libapriltag_amd_stop<n>.so drop every cluster after phase n (isaac_ros_apriltag_amd.build.build_amd_variant('stop<n>',
# ['AMDAT_FQ_STOP=<n>'])).  Run on the GPU box from the repo root; writes gpurun_out/<tag>/fq_phase_insts.md
TAG=${1:-fqphase}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for V in stop1 stop2 stop4 stop5 stop6 ""; do
  N=${V:-full}
  AMDAT_LIB=$V timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/p_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 32 1 8 > $OUT/p_$N.log 2>&1
  echo "### after phase: $N" >> $OUT/fq_phase_insts.md
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/p_$N k_fit_quads >> $OUT/fq_phase_insts.md 2>&1
  find $OUT/p_$N -name "*.db" -delete
done
cat $OUT/fq_phase_insts.md
