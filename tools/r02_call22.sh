#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02v
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_all.txt 2>&1
tail -6 $O/pytest_gpu_all.txt
timeout 600 python bench.py --steps 500 --warmup 50 > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | cut -c1-1200
timeout 300 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_v1_int8_b1.txt 2>&1
tail -4 $O/layers_mobilenet_v1_int8_b1.txt
