#!/bin/bash
# round 6, call 10: debugging the HSA dispatch stamps of the direct path (call 9: negative durations)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
TAMD_DEBUG=1 python tools/direct_timestamps.py mobilenet_v1 1 int8 3 2>&1 | grep -v "^Tengine" | grep "stamp\|packets per\|sum of" | head -50
