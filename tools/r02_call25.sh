#!/bin/bash
# where does a write-heavy pointwise layer's time go?  (1) raw store / load / mixed rates in the epilogues' access shapes,
# (2) PMC passes over pw_stream / pw_rows / igemm on 64 x 56^2 -> 256 at batch 32
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02y
mkdir -p $O
cd $R
timeout 120 tools/exp/write_bw.bin > $O/write_bw.txt 2>&1
cat $O/write_bw.txt
cd /tmp; export TMPDIR=/tmp
for m in pw_stream pw_rows igemm2; do
  for i in 1 2 3; do
    case $i in
     1) C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU";;
     2) C="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU";;
     3) C="TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum";;
    esac
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${m}_$i -- python $R/tools/run_layer.py 64 56 256 1 32 $m 5 > $O/pmc_${m}_$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $O/pmc_$m.csv $O/pmc_${m}_1 $O/pmc_${m}_2 $O/pmc_${m}_3 > /dev/null
  grep -E "^kernel|pw_|igemm" $O/pmc_$m.csv | cut -c1-900
  rm -rf $O/pmc_${m}_1 $O/pmc_${m}_2 $O/pmc_${m}_3
done
