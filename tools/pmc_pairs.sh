set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_ROUND:-r4}/${PMC_NAME:-pmc_pairs}
mkdir -p $OUT
cd $R
SPECS=${PMC_SPECS:-"1000:16:smsqfa 1000:16:sqfa 1000:32:smsqfa 1000:32:sqfa 1000:16:smsqfa:f64 1000:16:sqfa:f64"}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/a -o a --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 -d $OUT/b -o b --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $OUT/c -o c --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/s -o s --output-format csv -- python3 tools/run_pairs_once.py $SPECS > $OUT/s.log 2>&1
find $OUT -name "*.csv" | head -30
tail -3 $OUT/a.log
