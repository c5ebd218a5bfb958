#!/bin/bash
# uint8 division-free rounding + bench.py roofline.direct_dispatch: uint8 parity first, then tables, then the whole suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02ah
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_baseline_batches.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_uint8.txt
timeout 300 python tools/profile_layers.py mssd 16 10 uint8 > $O/layers_mssd_uint8_b16.txt 2>&1; grep -E "dw|sum of" $O/layers_mssd_uint8_b16.txt | head -16
timeout 300 python tools/profile_layers.py yolov3_tiny 8 10 uint8 > $O/layers_yolov3_tiny_uint8_b8.txt 2>&1; tail -1 $O/layers_yolov3_tiny_uint8_b8.txt
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err; tail -1 $O/bench_b1.json | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print(round(l['ms_per_step']*1e3,2),'us', round(l['value']),'img/s', json.dumps(l['roofline'])[:900])"
timeout 300 python bench.py --model mssd --dtype uint8 --batch 16 --steps 50 --no-cpu-baseline > $O/bench_mssd_uint8_b16.json 2>/dev/null; tail -1 $O/bench_mssd_uint8_b16.json | cut -c1-200
timeout 300 python bench.py --model yolov3_tiny --dtype uint8 --batch 8 --steps 50 --no-cpu-baseline > $O/bench_yolov3_tiny_uint8_b8.json 2>/dev/null; tail -1 $O/bench_yolov3_tiny_uint8_b8.json | cut -c1-200
timeout 200 python tools/tm_benchmark.py -r 100 -s 1 -p int8 2>&1 | tail -2
