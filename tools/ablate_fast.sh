# Timing of the fast GMM kernel with parts of its work compiled out (needs a build with
# -DAMLIKE_ABLATION_BUILD).  fablN bits: 1 no stores, 2 no exp-sum, 4 no segment closing,
# 8 no feature-fragment reads, 16 no barrier / staging.
for v in fast-f16x2:2x8 fabl1:2x8 fabl2:2x8 fabl4:2x8 fabl6:2x8 fabl8:2x8 fabl14:2x8 fabl30:2x8; do
  echo "$v"; timeout -k 10 200 python -u tools/perf_gmm.py swbd_tri4 35554 $v 2>&1 | grep "ms/launch"
done
