#!/bin/bash
# Collects what profiles/ holds for the current build, on the GPU box:
#   kernel stats of one bench run (rocprofv3 --kernel-trace --stats), the bench line itself,
#   PMC passes over the GMM scoring kernel, timings of the other configs and entry points.
# usage: tools/collect_profiles.sh <tag>      (outputs under gpurun_out/<tag>_*)
set -u
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${tag}_stats.log 2>&1 || { echo "stats run failed"; exit 1; }
cp $R/gpurun_out/${tag}_stats/*/*kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
cd $R
timeout -k 10 400 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "bench failed"; exit 1; }
timeout -k 10 600 bash tools/pmc_passes.sh ${tag} swbd_tri4 218541 diff-f16x2:2x8 > gpurun_out/${tag}_pmc.log 2>&1 || { echo "pmc failed"; exit 1; }
timeout -k 10 300 python3 -u tools/perf_configs.py > gpurun_out/${tag}_configs.txt 2>&1 || { echo "configs failed"; exit 1; }
timeout -k 10 300 python3 -u tools/perf_aux.py > gpurun_out/${tag}_aux.txt 2>&1 || { echo "aux failed"; exit 1; }
timeout -k 10 300 python3 -u tests/perf_pipeline.py > gpurun_out/${tag}_host_pipeline.txt 2>&1 || { echo "pipeline failed"; exit 1; }
echo done
