# Per-kernel timing of the nnet2 p-norm forward (one bench step set under rocprofv3).
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_n
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_n -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-to-host --steps 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_n.log 2>&1
grep -i "nnet" $GRAFT_REPO_ROOT/gpurun_out/prof_n/*/*kernel_stats.csv | cut -c1-60,150-260
python3 - <<PY
import csv,glob,os,json
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_n/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
for r in rows[-6:]:
    print(r.get("Grid_Size_X",""), r.get("Grid_Size_Y",""), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us")
line=[l for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_n.log") if l.startswith("{")]
if line:
    d=json.loads(line[-1]); print("nnet ms_per_step", d["config"]["nnet2_pnorm"]["ms_per_step"])
PY
