# A/B timing of library builds on one box: tools/ab.sh <lib.so> <lib.so> ... (paths relative to
# kaldi_amd/csrc); each is timed twice, interleaved.
for rep in 1 2; do
  for l in "$@"; do
    echo "$l"; AMLIKE_LIB_PATH=$PWD/kaldi_amd/csrc/$l timeout -k 10 200 python -u tools/perf_gmm.py ${AB_CONFIG:-swbd_tri4} ${AB_FRAMES:-35554} diff-f16x2:2x8 2>&1 | grep "ms/launch" || exit 1
  done
done
