#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02d
mkdir -p $O
cd $R
$R/tools/exp/pwdw_anatomy.bin > $O/pwdw_anatomy.txt 2>&1
cat $O/pwdw_anatomy.txt
timeout 900 python -m pytest tests/test_gpu_pwdw.py -q -x > $O/pytest_pwdw.txt 2>&1
tail -5 $O/pytest_pwdw.txt
timeout 300 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_v1_int8_b1.txt 2>&1
cat $O/layers_mobilenet_v1_int8_b1.txt
TAMD_FUSE_PWDW=2 timeout 300 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_v1_int8_b1_forced.txt 2>&1
tail -5 $O/layers_mobilenet_v1_int8_b1_forced.txt
