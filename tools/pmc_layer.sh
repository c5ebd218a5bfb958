#!/bin/bash
# Runs ON THE GPU BOX: three rocprofv3 --pmc passes (counters only, no tracing) over eager launches of ONE int8 conv layer with a
# pinned member of the GEMM family, merged into a per-kernel table.  usage: tools/pmc_layer.sh cin hw cout k batch member tag
CIN=$1; HW=$2; COUT=$3; K=$4; B=$5; MEMBER=$6; TAG=${7:-pmc_layer}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do
  case $i in
   1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY";;
   2) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE";;
   3) C="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC";;
  esac
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_$i -- python $R/tools/run_layer.py $CIN $HW $COUT $K $B "$MEMBER" 3 > $R/gpurun_out/${TAG}_$i.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}.csv $R/gpurun_out/${TAG}_1 $R/gpurun_out/${TAG}_2 $R/gpurun_out/${TAG}_3 > /dev/null
rm -rf $R/gpurun_out/${TAG}_1 $R/gpurun_out/${TAG}_2 $R/gpurun_out/${TAG}_3
grep -E "kernel,|pgemm|igemm" $R/gpurun_out/${TAG}.csv
