#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call19
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python tools/exp/dw_forms.py 11 12 14 21 22 2>&1 | grep -v "^Tengine" | tee $O/dw_forms.txt
