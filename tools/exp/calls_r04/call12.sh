#!/bin/bash
# round 4, call 12: where does a ResNet-50 b32 pass lose against the sum of its isolated launches?  Same plan, isolated table vs
# per-position bodies of hipGraph replays (rocprofv3 --kernel-trace), side by side
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_call12
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
export TAMD_PLAN_CACHE=$O/plan_resnet50_int8_b32.txt
timeout 300 python $R/tools/profile_layers.py resnet50 32 20 int8 2>&1 | grep -v "^Tengine" > $O/layers_resnet50_int8_b32.txt
tail -1 $O/layers_resnet50_int8_b32.txt
rm -rf $O/trace_is
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_is -- python $R/tools/replay_model.py resnet50 32 30 int8 > $O/replay.txt 2> $O/trace_is.err
T=$(find $O/trace_is -name "*kernel_trace.csv" | head -1)
N=$(grep -o "launches_per_replay [0-9]*" $O/replay.txt | cut -d' ' -f2)
python $R/tools/trace_gaps.py $T $N 30 > $O/insitu_trace_resnet50_int8_b32.txt 2>&1
python $R/tools/replay_model.py resnet50 32 30 int8 >> $O/insitu_trace_resnet50_int8_b32.txt 2>&1
rm -rf $O/trace_is
paste <(awk 'NR>1 && NF>=7 {printf "%-26s %-30s %8s\n", $1, $2, $3}' $O/layers_resnet50_int8_b32.txt) <(awk 'NR>1 && $1 ~ /^[0-9]+$/ {print $(NF-1)}' $O/insitu_trace_resnet50_int8_b32.txt) | awk '{d=$4-$3; printf "%s  %8.2f  %+7.2f\n", $0, $4, d}' > $O/isolated_vs_insitu.txt
cat $O/isolated_vs_insitu.txt
tail -3 $O/insitu_trace_resnet50_int8_b32.txt
