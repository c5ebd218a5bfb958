#!/bin/bash
# in-situ (inside the replayed hipGraph) duration of every launch vs the warm per-kernel table
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02q
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cfg in "resnet50 32 int8" "mobilenet_v1 64 int8"; do
  set -- $cfg
  rm -rf $O/trace
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/replay_model.py $1 $2 50 $3 > $O/replay_$1.txt 2> $O/trace.err
  tail -1 $O/replay_$1.txt
  T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  N=$(grep -o "launches_per_replay [0-9]*" $O/replay_$1.txt | cut -d' ' -f2)
  python $R/tools/trace_gaps.py $T $N 50 > $O/trace_gaps_$1_$3_b$2.txt 2>&1
  cat $O/trace_gaps_$1_$3_b$2.txt
  rm -rf $O/trace
done
