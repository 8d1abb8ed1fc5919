export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02d
mkdir -p $OUT
cd /tmp
for G in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32" "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-30)
  timeout 200 rocprofv3 --pmc $G -d $OUT/p_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/pipeline_once.py 32 1 8 > $OUT/p_$N.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/p_$N k_fit_quads k_points k_cc_local >> $OUT/pmc2.md 2>&1
  echo >> $OUT/pmc2.md
  find $OUT/p_$N -name "*.db" -delete
done
cat $OUT/pmc2.md
