export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02g
mkdir -p $OUT
cd /tmp
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC -d $OUT/p -o pmc -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 32 1 8 > $OUT/p.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/p k_fit_quads k_points k_cc_local k_threshold
