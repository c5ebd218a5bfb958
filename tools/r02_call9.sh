#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pwdw.py -q -x > $O/pytest_pwdw.txt 2>&1
tail -8 $O/pytest_pwdw.txt
timeout 300 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_v1_int8_b1.txt 2>&1
cat $O/layers_mobilenet_v1_int8_b1.txt
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_all.txt 2>&1
tail -8 $O/pytest_gpu_all.txt
