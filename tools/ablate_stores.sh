# Store-path experiment on the differenced GMM kernel (needs -DAMLIKE_ABLATION_BUILD):
# dabl1 = no output stores, dabl32 = the same number of store instructions, each writing 64
# consecutive floats instead of 64 different rows.
for v in diff-f16x2:2x8 dabl1:2x8 dabl32:2x8; do
  echo "$v"; timeout -k 10 200 python -u tools/perf_gmm.py swbd_tri4 35554 $v 2>&1 | grep "ms/launch"
done
