#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
for m in 2 1; do
  echo "== TAMD_CHAIN=$m"
  TAMD_CHAIN=$m timeout 200 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_b1_chain$m.txt 2>&1
  grep -i "chain\|sum of\|error\|fault" $O/layers_mobilenet_b1_chain$m.txt | cut -c1-200
done
TAMD_CHAIN=2 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "mobilenet" > $O/pytest_chain.txt 2>&1
tail -5 $O/pytest_chain.txt
