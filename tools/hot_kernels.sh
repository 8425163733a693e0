#!/bin/bash
# Kernel-by-kernel times of the eager hot-path step (GPU box): bash tools/hot_kernels.sh <tag> -> gpurun_out/<tag>_hot.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-hot}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph
rocprofv3 --kernel-trace --stats -f csv -d /tmp/ph -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full --no-graph > $R/gpurun_out/${T}_hot.log 2>&1
python3 $R/tools/prof_summary.py /tmp/ph 16 60 > $R/gpurun_out/${T}_hot.txt
