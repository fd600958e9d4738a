# Timing of the differenced GMM kernel with parts of its work compiled out (needs a build with
# -DAMLIKE_ABLATION_BUILD).  dablN bits: 2 no exp-sum, 4 no segment loop, 8 no feature-fragment
# reads, 16 no barrier / staging, 32 coalesced dummy stores.
for v in diff-f16x2:2x8 dabl1:2x8 dabl9:2x8 dabl2:2x8 dabl4:2x8 dabl6:2x8 dabl8:2x8 dabl14:2x8 dabl30:2x8 dabl32:2x8; do
  echo "$v"; timeout -k 10 200 python -u tools/perf_gmm.py swbd_tri4 35554 $v 2>&1 | grep "ms/launch"
done
