#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call39
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_softmax_i8.py tests/test_gpu_edge_cases.py tests/test_plugin_dropin.py -q -m gpu --tb=short -p no:cacheprovider -k "softmax or unsupported or mixed_graph or resnet50_prob or resblock_tail or wrong_input" > $O/pytest.txt 2>&1
grep -E "passed|failed|error" $O/pytest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest.txt | head -12
timeout 120 python tools/exp/softmax_i8_resnet50.py 32 > $O/softmax_i8_resnet50_b32.txt 2>&1
tail -4 $O/softmax_i8_resnet50_b32.txt
