#!/bin/bash
# round 6, call 19: why is the matrix-core depthwise body faster in the harness (memset operands) and slower in the graph?  the harness with random operands
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call19
mkdir -p $O
cd $R
(for rnd in 0 1; do for mm in 1 0; do echo "== random=$rnd mm=$mm"; if [ $rnd = 1 ]; then export DWPW_RANDOM=1; else unset DWPW_RANDOM; fi; TAMD_PIN=dwpw_mm=$mm timeout 200 tools/exp/dwpw_anatomy.bin | sed -n 3,5p; done; done) 2>&1 | tee $O/dwpw_anatomy_random.txt
