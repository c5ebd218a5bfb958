#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02u
mkdir -p $O
cd $R
timeout 120 tools/exp/chain_anatomy.bin > $O/chain_anatomy.txt 2>&1
cat $O/chain_anatomy.txt
