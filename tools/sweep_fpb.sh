# Frames per block of the GMM scoring kernel: AMLIKE_GMM_FPB sweep (unset = the launch code's own choice).
T=${AB_FRAMES:-218541}
for rep in 1 2; do
  echo "FPB default"; timeout -k 10 200 python -u tools/perf_gmm.py swbd_tri4 $T diff-f16x2:2x8 2>&1 | grep "ms/launch" || exit 1
  for f in 1024 1536 2048 3072 4096 6144 8192; do
    echo "FPB $f"; AMLIKE_GMM_FPB=$f timeout -k 10 200 python -u tools/perf_gmm.py swbd_tri4 $T diff-f16x2:2x8 2>&1 | grep "ms/launch" || exit 1
  done
done
