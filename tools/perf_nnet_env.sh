# Per-kernel times of the nnet2 forward (bench workload) under different environment settings, on one
# box: tools/perf_nnet_env.sh "NAME=value ..." "NAME=value ..." (an empty string = defaults).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for setting in "$@"; do
  rm -rf $R/gpurun_out/prof_env
  ( for kv in $setting; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_env -- python3 $R/bench.py --no-cpu-baseline --no-host-to-host --steps 3 > $R/gpurun_out/prof_env.log 2>&1 ) || exit 1
  echo "== [$setting]"
  python3 - <<PY
import csv,glob,os,json
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/prof_env/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "nnet" in r["Name"]:
        print("%-72s calls %4s avg %9.1f us" % (r["Name"][13:85], r["Calls"], float(r["AverageNs"])/1e3))
line=[l for l in open(R+"/gpurun_out/prof_env.log") if l.startswith("{")]
if line:
    d=json.loads(line[-1]); print("nnet ms_per_step %.3f   gmm kernel_ms %.3f" % (d["config"]["nnet2_pnorm"]["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
done
