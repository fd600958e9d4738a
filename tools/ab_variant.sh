# Times GMM kernel variants "<mode>:<NT>x<W>" on one box, twice each.
for rep in 1 2; do
  for v in "$@"; do
    echo "$v"; timeout -k 10 200 python -u tools/perf_gmm.py ${AB_CONFIG:-swbd_tri4} ${AB_FRAMES:-35554} $v 2>&1 | grep "ms/launch" || exit 1
  done
done
