# Per-kernel times of the nnet2 forward for several library builds (A/B on one box).
# AB_ARGS: bench.py arguments that pick the workload (default: the small rm_tri1 one).
cd /tmp && export TMPDIR=/tmp
for l in "$@"; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ab
  AMLIKE_LIB_PATH=$GRAFT_REPO_ROOT/kaldi_amd/csrc/$l timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ab -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-to-host --steps 3 ${AB_ARGS:---config rm_tri1 --utts-per-gpu 40} > $GRAFT_REPO_ROOT/gpurun_out/prof_ab.log 2>&1 || exit 1
  echo "== $l"
  python3 - <<PY
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_ab/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "nnet" in r["Name"]:
        print("%-64s calls %4s avg %9.1f us" % (r["Name"][13:77], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
