# Average shader clock during the GMM scoring kernel, on real and on all-zero features:
# GRBM_GUI_ACTIVE (busy cycles of the dispatch, summed over the 8 XCDs) / 8 / the dispatch's duration.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in real zero; do
  rm -rf $R/gpurun_out/clk_$f
  PERF_FEATS=$f timeout -s KILL 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clk_$f -- python3 $R/tools/perf_gmm.py swbd_tri4 218541 diff-f16x2:2x8 > $R/gpurun_out/clk_$f.log 2>&1 || { echo "pass $f failed"; exit 1; }
done
python3 - <<'P'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
for f in ("real", "zero"):
    files = glob.glob(R + "/gpurun_out/clk_%s/**/*counter_collection.csv" % f, recursive=True)
    rows = [r for r in csv.DictReader(open(files[0])) if "gmm_score_diff_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    ghz = []
    for r in rows:
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        ghz.append(float(r["Counter_Value"]) / 8.0 / dur)
    print("%s features: %d dispatches, kernel %.2f ms, GRBM_GUI_ACTIVE / 8 XCDs / duration = %.2f GHz" % (
        f, len(rows), sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows) / len(rows) / 1e6, sum(ghz) / len(ghz)))
P
