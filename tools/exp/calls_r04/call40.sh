#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call40
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_gpu_softmax_i8.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_softmax.txt 2>&1
grep -E "passed|failed|error" $O/pytest_softmax.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest_softmax.txt | head -12
timeout 40 python tools/exp/softmax_i8_resnet50.py 32 > $O/softmax_i8_resnet50_b32.txt 2>&1
tail -4 $O/softmax_i8_resnet50_b32.txt
timeout 60 python -m pytest tests/test_gpu_glue_int8.py tests/test_gpu_edge_cases.py tests/test_plugin_dropin.py -q -m gpu --tb=short -p no:cacheprovider -k "glue or relu or eltwise or concat or pool or unsupported or mixed_graph or resnet50_prob" > $O/pytest_rest.txt 2>&1
grep -E "passed|failed|error" $O/pytest_rest.txt | tail -3
grep -E "^FAILED|^ERROR|differ|^E  " $O/pytest_rest.txt | head -12
