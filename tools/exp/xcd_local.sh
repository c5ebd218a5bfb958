#!/bin/bash
# Does the batch-1 chain get faster when it runs on ONE XCD (activations handed over through that XCD's L2 instead of the memory
# side)?  Timing only: the no-fence variants are not byte-safe (see graph.hip exp_plain_kernels).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04_xcd}; mkdir -p $O; cd $R
tools/exp/cumask_probe.bin | tee $O/cumask_probe.txt
timeout 900 python tools/exp/ab_step.py mobilenet_v1 1 int8 500 3 \
   "default(coherent,all_CUs)" \
   "coherent,first32=TAMD_DIRECT_CU_MASK=first32" "coherent,stride8=TAMD_DIRECT_CU_MASK=stride8" \
   "plain+agent_fences,all=TAMD_EXP_PLAIN_KERNELS=1" \
   "plain+nofence,all=TAMD_EXP_PLAIN_KERNELS=1,TAMD_EXP_NOFENCE=1" \
   "plain+nofence,first32=TAMD_EXP_PLAIN_KERNELS=1,TAMD_EXP_NOFENCE=1,TAMD_DIRECT_CU_MASK=first32" \
   "plain+nofence,stride8=TAMD_EXP_PLAIN_KERNELS=1,TAMD_EXP_NOFENCE=1,TAMD_DIRECT_CU_MASK=stride8" \
   "plain+nofence,first64=TAMD_EXP_PLAIN_KERNELS=1,TAMD_EXP_NOFENCE=1,TAMD_DIRECT_CU_MASK=first64" \
   "plain+nofence,stride4=TAMD_EXP_PLAIN_KERNELS=1,TAMD_EXP_NOFENCE=1,TAMD_DIRECT_CU_MASK=stride4" \
   2>&1 | grep -v "^Tengine" | tee $O/xcd_local_b1.txt
