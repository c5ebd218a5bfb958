#!/bin/bash
# one-FMA requantisation + two-FMA residual tail + MFMA VGPR form + Winograd F(2,3): parity first, then the tables
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02w
mkdir -p $O
cd $R
timeout 60 tools/exp/cvt_pk_u8_probe.bin > $O/cvt_pk_u8_probe.txt 2>&1
cat $O/cvt_pk_u8_probe.txt | head -30
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_all.txt 2>&1
tail -8 $O/pytest_gpu_all.txt
timeout 300 python tools/profile_layers.py resnet50 32 20 int8 > $O/layers_resnet50_int8_b32.txt 2>&1
tail -3 $O/layers_resnet50_int8_b32.txt
timeout 300 python tools/profile_layers.py mobilenet_v1 64 20 int8 > $O/layers_mobilenet_v1_int8_b64.txt 2>&1
tail -3 $O/layers_mobilenet_v1_int8_b64.txt
timeout 300 python tools/profile_layers.py mobilenet_v1 1 50 int8 > $O/layers_mobilenet_v1_int8_b1.txt 2>&1
tail -2 $O/layers_mobilenet_v1_int8_b1.txt
timeout 300 python tools/profile_layers.py squeezenet_v1.1 1 50 fp32 > $O/layers_squeezenet_fp32_b1.txt 2>&1
tail -2 $O/layers_squeezenet_fp32_b1.txt
timeout 600 python bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | cut -c1-900
