#!/bin/bash
# round 4, first GPU call: (1) the new / changed GPU tests, (2) A/B of the allocation scheme on the batched configs,
# (3) the blocking run's variants with its host-side anatomy, (4) fp32 Winograd race log, (5) the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call1
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_async.py tests/test_gpu_plan_cache.py tests/test_gpu_direct.py tests/test_plugin_dropin.py tests/test_gpu_parity_fp32.py -x -q -m gpu -s > $O/pytest_subset.txt 2>&1
tail -5 $O/pytest_subset.txt
# ---- (2) allocation A/B, same plan, interleaved
for cfg in "resnet50 32 int8 30" "mobilenet_v1 64 int8 50" "mobilenet_v1 1 int8 500"; do
  set -- $cfg
  timeout 600 python tools/exp/ab_step.py $1 $2 $3 $4 5 "separate_hipMallocs=TAMD_ARENA=0,TAMD_POOL=0" "arena=TAMD_POOL=0" "arena+shared_buffers" "arena+shared_buffers(2)" 2>&1 | grep -v "^Tengine" | tee -a $O/ab_alloc.txt
done
for cfg in "yolov3_tiny 8 uint8 30" "mssd 16 uint8 30"; do
  set -- $cfg
  timeout 600 python tools/exp/ab_step.py $1 $2 $3 $4 5 "separate_hipMallocs=TAMD_ARENA=0" "arena" "arena(2)" 2>&1 | grep -v "^Tengine" | tee -a $O/ab_alloc.txt
done
# ---- (3) blocking run variants
timeout 600 python tools/exp/h2h_variants.py mobilenet_v1 1 1000 5 > $O/h2h_variants.txt 2>&1
cat $O/h2h_variants.txt | tail -20
# ---- (4) Winograd race, fp32 at batch 8
for m in resnet50 yolov3_tiny; do
  TAMD_DEBUG=1 timeout 300 python tools/run_model.py $m 8 1 fp32 2>&1 | grep -i "winograd" > $O/f32_winograd_race_${m}_b8.txt
  echo "$m: $(grep -c 'winograd$' $O/f32_winograd_race_${m}_b8.txt) of $(wc -l < $O/f32_winograd_race_${m}_b8.txt) layers pick winograd"
done
# ---- (5) the default bench line (with the CPU baseline under the default OpenMP cap)
timeout 900 python bench.py > $O/bench_b1_default.json 2> $O/bench_b1_default.err
tail -1 $O/bench_b1_default.json | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('value %.0f  ms/step %.4f  h2h %s pipelined %s' % (j['value'], j['ms_per_step'], j.get('host_to_host_images_per_s'), j.get('host_to_host_pipelined_images_per_s')))
c = j['cpu_baseline']; print('cpu', c['value'], c['cores'], c.get('openmp_cap'), [(p['threads'], round(p['min_ms'],1)) for p in c['sweep']])
print('roofline', j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['avg_launch_us'])
"
tail -3 $O/bench_b1_default.err
