#!/bin/bash
# Collect the bench line, the rocprofv3 kernel stats and the two PMC traffic passes of one bench command (GPU box).
# usage: bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc_traffic.json}
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r02_v5}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "bench done"
rm -rf /tmp/ps /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats -f csv -d /tmp/ps -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full > $O/${T}_stats.log 2>&1
cp /tmp/ps/*/*_kernel_stats.csv $O/${T}_kernel_stats.csv
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d /tmp/pf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full > $O/${T}_pf.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d /tmp/pw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full > $O/${T}_pw.log 2>&1
echo "write done"
python3 $R/tools/pmc_traffic.py /tmp/pf/*/*counter_collection.csv /tmp/pw/*/*counter_collection.csv $O/${T}_pmc_traffic.json
rm -rf /tmp/pm /tmp/pg
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -f csv -d /tmp/pm -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full > $O/${T}_pm.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -f csv -d /tmp/pg -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full > $O/${T}_pg.log 2>&1
python3 $R/tools/pmc_mfma.py /tmp/pm/*/*counter_collection.csv /tmp/pg/*/*counter_collection.csv $O/${T}_pmc_mfma.json
echo "mfma utilisation done"
rm -rf /tmp/pfull
rocprofv3 --kernel-trace --stats -f csv -d /tmp/pfull -- python3 $R/tools/full_profile.py 5 2 > $O/${T}_full.log 2>&1
cp /tmp/pfull/*/*_kernel_stats.csv $O/${T}_full_workload_kernel_stats.csv
echo "full workload stats done"
rm -rf /tmp/pff /tmp/pfw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d /tmp/pff -- python3 $R/tools/full_profile.py 2 1 > $O/${T}_full_pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d /tmp/pfw -- python3 $R/tools/full_profile.py 2 1 > $O/${T}_full_pw.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pff/*/*counter_collection.csv /tmp/pfw/*/*counter_collection.csv $O/${T}_full_pmc_traffic.json
echo "full workload traffic done"
