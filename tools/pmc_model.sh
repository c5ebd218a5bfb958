#!/bin/bash
# Runs ON THE GPU BOX: three rocprofv3 --pmc passes (counters only, no tracing) over an eager run of a config model,
# merged into a per-kernel table.  usage: tools/pmc_model.sh <model> <batch> <int8|uint8|fp32> <tag>
M=${1:-mobilenet_v1}; B=${2:-64}; D=${3:-int8}; TAG=${4:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do
  case $i in
   1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY";;
   2) C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE";;
   3) C="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC";;
  esac
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_$i -- python $R/tools/run_model.py $M $B 2 $D > $R/gpurun_out/${TAG}_$i.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}.csv $R/gpurun_out/${TAG}_1 $R/gpurun_out/${TAG}_2 $R/gpurun_out/${TAG}_3 > /dev/null
wc -l $R/gpurun_out/${TAG}.csv
