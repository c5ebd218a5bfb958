#!/bin/bash
# round 6, call 16: dwpw -- what the stage skeleton consists of (ablations of the A-fragment stream, the tap loads, the B reads), and the
# committed LDS-constants form against the scalar-base addressing form on MobileNet-v1 b64
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call16
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 200 tools/exp/dwpw_anatomy.bin 2>&1 | tee $O/dwpw_anatomy.txt
AB_LAYERS=1 timeout 1500 python tools/exp/ab_lib.py mobilenet_v1 64 int8 200 4 ldsconsts=tools/exp/ab/ldsconsts.so saddr=product 2>&1 | grep -v "^Tengine" | tee $O/ab_mobilenet_b64_saddr.txt | head -12
