set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG:-r1k}_stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG:-r1k}_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG:-r1k}_stats.log 2>&1 || { echo "stats failed"; exit 1; }
cp $R/gpurun_out/${TAG:-r1k}_stats/*/*kernel_stats.csv $R/gpurun_out/${TAG:-r1k}_kernel_stats.csv
cd $R
timeout -k 10 600 python3 bench.py > gpurun_out/${TAG:-r1k}_bench.json 2> gpurun_out/${TAG:-r1k}_bench.err || { echo "bench failed"; exit 1; }
echo done
