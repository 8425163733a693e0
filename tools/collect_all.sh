#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call:  bash tools/collect_all.sh   (from the repo root)
# bench line + rocprofv3 stats + PMC traffic (collect_profiles.sh), cfg5 stats, eager / reference-shape bench lines,
# kNN robustness cases, segment-diameter bench (PMC: HBM traffic and matrix-pipe busy cycles, separate passes).  Copy gpurun_out/<tag>_* into profiles/ afterwards.
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
T=r03_v4
bash tools/collect_profiles.sh $T > gpurun_out/${T}_collect.log 2>&1 && \
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc5 && \
rocprofv3 --kernel-trace --stats -f csv -d /tmp/pc5 -- python3 $GRAFT_REPO_ROOT/tools/cfg5_profile.py 5 2 > $GRAFT_REPO_ROOT/gpurun_out/${T}_cfg5.log 2>&1 && \
cp /tmp/pc5/*/*_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/${T}_cfg5_kernel_stats.csv && \
cd $GRAFT_REPO_ROOT && \
python3 bench.py --no-graph --no-cpu-baseline --no-full > gpurun_out/${T}_bench_eager.json 2>> gpurun_out/${T}_bench.err && \
python3 bench.py --batch 3 --points 7000 --k 80 --no-cpu-baseline --no-full > gpurun_out/${T}_bench_refshape.json 2>> gpurun_out/${T}_bench.err && \
python3 tools/knn_fallback_bench.py > gpurun_out/${T}_knn_fallback.log 2>&1 && \
python3 tools/segdiam_bench.py > gpurun_out/${T}_segdiam_bench.log 2>&1 && \
( echo "# forward_grouping, fused device path vs the literal per-(cloud, class) path (tools/grouping_bench.py B N P reps)"; \
  echo "## BASELINE configs[3] shape, 1 x 100000 points, 10 classes, 40 sheet-like blobs, reference thresholds (0.989, 0.0)"; python3 tools/grouping_bench.py 1 100000 10 3; \
  echo "## 1 x 100000 uniform cube (SURVEY 8d config 4)"; python3 tools/grouping_bench.py 1 100000 10 3 uniform; \
  echo "## 8 x 8192"; python3 tools/grouping_bench.py 8 8192 10 3 ) > gpurun_out/${T}_grouping_bench.log 2>&1 && \
tail -3 gpurun_out/${T}_collect.log && cut -c1-200 gpurun_out/${T}_bench.json
