# One bench line per GMM config (device-resident rate, roofline, the reference CPU rate on one
# thread and on all cores beside it), then the default line (swbd tri4 + the nnet2 leg).
set -u
R=$GRAFT_REPO_ROOT
cd $R
: > gpurun_out/r1k_configs.jsonl
for c in yesno_mono rm_tri1 wsj_tri4b; do
  timeout -k 10 300 python3 bench.py --config $c --no-nnet >> gpurun_out/r1k_configs.jsonl 2> gpurun_out/r1k_$c.err || { echo "bench $c failed"; exit 1; }
done
timeout -k 10 600 python3 bench.py > gpurun_out/r1k_bench.json 2> gpurun_out/r1k_bench.err || { echo "bench failed"; exit 1; }
cat gpurun_out/r1k_bench.json >> gpurun_out/r1k_configs.jsonl
echo done
