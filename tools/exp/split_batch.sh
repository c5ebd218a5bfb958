#!/bin/bash
# Does a per-GPU batch run faster as several concurrent sub-batch graphs (their launch boundaries and tile tails overlap)?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r04_split}; mkdir -p $O; cd $R
run() { # model dtype batch streams
  timeout 600 python bench.py --model $1 --dtype $2 --batch $3 --streams $4 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-14s %-6s batch %3d x %d streams: %9.0f img/s  %8.3f ms/step' % ('$1', '$2', $3, $4, j['value'], j['ms_per_step']))"
}
{
run resnet50 int8 32 1; run resnet50 int8 16 2; run resnet50 int8 8 4
run mobilenet_v1 int8 64 1; run mobilenet_v1 int8 32 2; run mobilenet_v1 int8 16 4
run yolov3_tiny uint8 8 1; run yolov3_tiny uint8 4 2
TAMD_U8_INT=1 run yolov3_tiny uint8 8 1; TAMD_U8_INT=1 run yolov3_tiny uint8 4 2
run mssd uint8 16 1; run mssd uint8 8 2
} 2>&1 | tee $O/split_batch.txt
