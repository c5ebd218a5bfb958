#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ag
mkdir -p $O
cd $R
timeout 200 python tools/direct_timeline.py mobilenet_v1 1 int8 30 > $O/direct_timeline_mobilenet_v1_int8_b1.txt 2>&1; cat $O/direct_timeline_mobilenet_v1_int8_b1.txt
TAMD_DIRECT_DISPATCH=1 timeout 900 python -m pytest tests/test_plugin_dropin.py tests/test_tm_benchmark_hip.py tests/test_gpu_parity.py tests/test_gpu_parity_uint8.py tests/test_gpu_parity_fp32.py -q -x -m gpu 2>&1 | tail -5 | tee $O/pytest_env_direct.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_all.txt 2>&1
tail -4 $O/pytest_gpu_all.txt
