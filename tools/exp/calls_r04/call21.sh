#!/bin/bash
# round 4, call 21: tile configurations of the fused pointwise+depthwise launches at batch 64 (every pair pinned to one configuration)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call21
mkdir -p $O
cd $R
export TMPDIR=/tmp
for cfg in "" "7,14,256" "7,7,256" "14,14,256" "4,14,256" "7,28,256" "14,28,512" "8,8,256" "4,7,256" "14,14,512"; do
  echo "== TAMD_PWDW_CFG=$cfg"
  TAMD_FUSE_PWDW=2 TAMD_PWDW_CFG=$cfg timeout 300 python tools/profile_layers.py mobilenet_v1 64 10 int8 2>&1 | grep -E "dw_i8|pwdw|sum of" | awk '{printf "%-28s %-30s %8s\n", $1,$2,$3}'
done | tee $O/pwdw_cfgs_b64.txt
