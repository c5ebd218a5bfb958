#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity_uint8.py tests/test_gpu_parity_fp32.py tests/test_plugin_dropin.py -q -x -k "priorbox or any_axis or detection_output or mssd_full or mssd_tail" > $O/pytest_priorbox.txt 2>&1
tail -15 $O/pytest_priorbox.txt
cd /tmp; export TMPDIR=/tmp
rm -rf $O/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/replay_model.py resnet50 32 30 int8 > $O/replay_resnet50.txt 2> $O/trace_resnet.err
tail -1 $O/replay_resnet50.txt; tail -3 $O/trace_resnet.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
N=$(grep -o "launches_per_replay [0-9]*" $O/replay_resnet50.txt | cut -d' ' -f2)
[ -n "$T" ] && python $R/tools/trace_gaps.py $T $N 30 > $O/trace_gaps_resnet50_int8_b32.txt 2>&1
cat $O/trace_gaps_resnet50_int8_b32.txt
rm -rf $O/trace
