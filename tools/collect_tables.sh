#!/bin/bash
# Runs ON THE GPU BOX: per-launch tables (HIP events, tamd_graph_profile) of the BASELINE configs at the per-GPU batch, the
# side bench lines and tm_benchmark's own loop.  usage: tools/collect_tables.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python tools/profile_layers.py mobilenet_v1 1 50 int8  > $O/layers_mobilenet_v1_int8_b1.txt 2>&1
python tools/profile_layers.py mobilenet_v1 64 10 int8 > $O/layers_mobilenet_v1_int8_b64.txt 2>&1
python tools/profile_layers.py resnet50 32 5 int8      > $O/layers_resnet50_int8_b32.txt 2>&1
python tools/profile_layers.py yolov3_tiny 8 10 uint8  > $O/layers_yolov3_tiny_uint8_b8.txt 2>&1
python tools/profile_layers.py mssd 16 10 uint8        > $O/layers_mssd_uint8_b16.txt 2>&1
python tools/profile_layers.py squeezenet_v1.1 1 20 fp32 > $O/layers_squeezenet_fp32_b1.txt 2>&1
python bench.py --model yolov3_tiny --dtype uint8 --batch 8 --steps 50 --cpu-seconds 6 > $O/bench_yolov3_tiny_uint8_b8.json 2> $O/bench_yolo.err
python bench.py --model mssd --dtype uint8 --batch 16 --steps 50 --cpu-seconds 6 > $O/bench_mssd_uint8_b16.json 2> $O/bench_mssd.err
python bench.py --model resnet50 --batch 32 --steps 30 --cpu-seconds 6 > $O/bench_resnet50_int8_b32.json 2> $O/bench_rn.err
python bench.py --model mobilenet_v1 --batch 64 --steps 50 --no-cpu-baseline > $O/bench_mobilenet_v1_int8_b64.json 2> $O/bench_mb64.err
python bench.py --streams 16 --steps 2000 --no-cpu-baseline > $O/bench_mobilenet_b1_16streams.json 2> $O/bench_s16.err
(python tools/tm_benchmark.py -r 100 -s 1 -p int8; python tools/tm_benchmark.py -r 30 -s 5 -p int8 -b 32; python tools/tm_benchmark.py -r 30 -s 8 -p uint8 -b 8) > $O/tm_benchmark_host_to_host.txt 2>&1
for f in $O/layers_*.txt; do echo $f; tail -1 $f; done
for f in $O/bench_*.json; do tail -1 $f | cut -c1-200; done
cat $O/tm_benchmark_host_to_host.txt | tail -4
