#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_async.py tests/test_plugin_dropin.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -q -x > $O/pytest_io.txt 2>&1
tail -6 $O/pytest_io.txt
timeout 600 python bench.py --steps 500 --warmup 50 > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['host_to_host'])"
timeout 600 python bench.py --model resnet50 --batch 32 --steps 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resnet50 b32', l['value'], l['ms_per_step'], l['host_to_host'])"
