#!/bin/bash
# round 4, call 34: rocprofv3 --kernel-trace --stats of the side bench commands (shipped plans: no autotune launches in the CSVs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call34
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {   # name, bench args...
  name=$1; shift
  rm -rf $O/trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py "$@" --no-cpu-baseline > $O/bench_${name}_under_rocprofv3.json 2> $O/trace_$name.err
  find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_bench_$name.csv
  echo "== $name"; head -6 $O/rocprofv3_kernel_stats_bench_$name.csv | cut -c1-160
}
run resnet50_int8_b32 --model resnet50 --batch 32 --steps 200 --warmup 10
run yolov3_tiny_uint8_b8 --model yolov3_tiny --dtype uint8 --batch 8 --steps 100 --warmup 5
run mssd_uint8_b16 --model mssd --dtype uint8 --batch 16 --steps 100 --warmup 5
run mobilenet_v1_int8_b64 --model mobilenet_v1 --batch 64 --steps 200 --warmup 10
rm -rf $O/trace
