#!/bin/bash
# round 6, call 40: which packets of a pair's stamped passes carry no stamps in a pytest process (diagnostic message), and how many parts pay
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_call40
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest "tests/test_gpu_split_batch.py::test_direct_timestamps_of_a_pair_list_both_halves" -m gpu -q --tb=short -s 2>&1 | grep -v "^Tengine" | grep "^E \|passed\|failed\|pair, HSA" | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_split_batch.py -m gpu -q --tb=short -s 2>&1 | grep -v "^Tengine" | grep "^E \|passed\|failed\|pair, HSA" | cut -c1-400
for cfg in "mobilenet_v1 64" "resnet50 32"; do
  timeout 900 python tools/exp/split_parts.py $cfg 100 5 2>&1 | grep -v "^Tengine" | tail -1
done | tee $O/split_parts.txt
