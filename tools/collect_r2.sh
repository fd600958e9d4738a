#!/bin/bash
# One set of evidence for the default bench workload (run on the GPU box from the repo root):
#   <tag>_kernel_stats.csv / _kernel_trace_head.csv   rocprofv3 --kernel-trace --stats of bench.py
#   <tag>_bench.json                                   the line an un-profiled bench.py prints
#   pmc_<tag>_summary.txt                              counter passes over the GMM scoring kernel
#   pmc_<tag>n_summary.txt                             counter passes over the nnet2 forward kernels
#   <tag>_clock.txt                                    shader clock under load, real and zero features
# usage: tools/collect_r2.sh <tag>
set -u
R=$GRAFT_REPO_ROOT
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --no-cpu-baseline --no-host-to-host > $R/gpurun_out/${tag}_stats.log 2>&1 || { echo "stats failed"; exit 1; }
cp $R/gpurun_out/${tag}_stats/*/*kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
cd $R
echo "stats done"
timeout -k 10 600 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "bench failed"; exit 1; }
echo "bench done"
bash tools/pmc_passes.sh ${tag} swbd_tri4 218541 diff-f16x2:2x8 > /dev/null 2>&1
echo "gmm pmc done"
# nnet2: the traffic counters too
cd /tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}n_${name} -- python3 $R/bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > $R/gpurun_out/pmc_${tag}n_${name}.log 2>&1 || echo "pass $name failed"
done
cd $R
bash tools/pmc_nnet.sh ${tag}n > /dev/null 2>&1
echo "nnet pmc done"
bash tools/clock_under_load.sh > gpurun_out/${tag}_clock.txt 2>&1
cat gpurun_out/${tag}_clock.txt
