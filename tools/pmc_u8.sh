cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  case $i in
   1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY";;
   2) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE";;
   3) C="SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_MISC";;
  esac
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_u8_$i -- python $R/tools/run_model.py yolov3_tiny 8 2 uint8 > $R/gpurun_out/pmc_u8_$i.log 2>&1
  tail -2 $R/gpurun_out/pmc_u8_$i.log
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_u8_b8.csv $R/gpurun_out/pmc_u8_1 $R/gpurun_out/pmc_u8_2 $R/gpurun_out/pmc_u8_3 | cut -c1-100 | head -5
