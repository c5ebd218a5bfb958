#!/bin/bash
# round-2 evidence after the requant / direct-dispatch work: bench + rocprofv3 + PMC traffic, tables of every config
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 bash tools/collect_profiles.sh r02fin 2>&1 | tail -45
timeout 900 bash tools/collect_tables.sh r02fin 2>&1 | tail -30
