#!/bin/bash
# round 4, call 18: forms of the int8 depthwise kernel at batch 64 (one layer, isolated launches)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call18
mkdir -p $O
cd $R
export TMPDIR=/tmp
for form in 12 14; do
  echo "== TAMD_DW_FORM=$form"
  TAMD_DW_FORM=$form timeout 300 python tools/profile_layers.py mobilenet_v1 64 20 int8 2>&1 | grep -E "/dw " | awk '{printf "%-28s %-30s %8s\n", $1,$2,$3}'
done | tee $O/dw_forms_b64.txt
