#!/bin/bash
# Runs ON THE GPU BOX (gpurun), round 2 call 1: where a batch-1 launch's time goes + parity at the BASELINE batches.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
$R/tools/exp/launch_chain.bin 30 300 > $O/launch_chain.txt 2>&1
cat $O/launch_chain.txt
timeout 900 python -m pytest tests/test_gpu_baseline_batches.py -x -q > $O/pytest_baseline_batches.txt 2>&1
tail -5 $O/pytest_baseline_batches.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/replay_model.py mobilenet_v1 1 200 int8 > $O/replay.txt 2> $O/trace.err
tail -1 $O/replay.txt
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
N=$(grep -o "launches_per_replay [0-9]*" $O/replay.txt | cut -d' ' -f2)
python $R/tools/trace_gaps.py $T $N 200 > $O/trace_gaps_mobilenet_v1_int8_b1.txt 2>&1
cat $O/trace_gaps_mobilenet_v1_int8_b1.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cd $R
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_b1.json 2> $O/bench_b1.err
tail -1 $O/bench_b1.json | cut -c1-1500
