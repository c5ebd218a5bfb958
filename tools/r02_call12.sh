#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_glue_int8.py tests/test_gpu_async.py tests/test_plugin_dropin.py -q -x > $O/pytest_new.txt 2>&1
tail -15 $O/pytest_new.txt
timeout 600 python bench.py --steps 500 --warmup 50 > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['value'], l['ms_per_step'], l['roofline'], l['host_to_host'])"
TG_DEBUG_TIME=1 timeout 120 python tools/run_model.py mobilenet_v1 1 1 int8 2>&1 | tail -20
