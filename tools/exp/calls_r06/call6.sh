#!/bin/bash
# round 6, call 6: fc7 folded into the launch in front of it by atomics as a probe; then the evidence pass of the round
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/fc_fold_probe.bin tools/exp/fc_fold_probe.hip 2>&1 | tail -3
/tmp/fc_fold_probe.bin 2>&1 | tee $O/fc_fold_probe.txt
bash tools/collect_evidence_r06.sh r06 2>&1 | tee $O/collect.log | tail -150
