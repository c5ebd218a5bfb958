#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pwdw.py -q -x > $O/pytest_pwdw.txt 2>&1
tail -15 $O/pytest_pwdw.txt
timeout 600 python -m pytest tests/test_gpu_gemm_family.py -q -x -k "pw_small" > $O/pytest_pw_small.txt 2>&1
tail -5 $O/pytest_pw_small.txt
timeout 300 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_v1_int8_b1.txt 2>&1
cat $O/layers_mobilenet_v1_int8_b1.txt
timeout 600 python bench.py --steps 500 --warmup 50 > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | cut -c1-900
