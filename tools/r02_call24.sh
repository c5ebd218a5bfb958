#!/bin/bash
# pw_rows (row-major persistent pointwise kernel): parity of the whole suite, member timings on the shallow-K shapes, model tables
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02x
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_all.txt 2>&1
tail -8 $O/pytest_gpu_all.txt
export BENCH_MEMBERS=igemm0,igemm2,igemm5,pw_stream,pw_rows,conv_igemm2
for bpc in 0 1 3; do
  TAMD_PW_ROWS_BPC=$bpc timeout 300 python tools/bench_members.py pw 32 > $O/members_pw_b32_bpc$bpc.txt 2>&1
  tail -8 $O/members_pw_b32_bpc$bpc.txt
done
unset BENCH_MEMBERS
timeout 300 python tools/profile_layers.py resnet50 32 20 int8 > $O/layers_resnet50_int8_b32.txt 2>&1
grep -E "branch2c|branch1|sum of" $O/layers_resnet50_int8_b32.txt | head -24
timeout 300 python tools/profile_layers.py mobilenet_v1 64 20 int8 > $O/layers_mobilenet_v1_int8_b64.txt 2>&1
cat $O/layers_mobilenet_v1_int8_b64.txt
timeout 600 python bench.py --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | cut -c1-300
