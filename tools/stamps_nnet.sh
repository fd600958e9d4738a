# Clock ticks per W tile of the fused output-layer + softmax kernel, for library builds made with
# -DAMLIKE_OUT_STAMPS (tools/build_variant.sh): tools/stamps_nnet.sh <lib.so> ...
for l in "$@"; do
  echo "== $l"
  AMLIKE_LIB_PATH=$GRAFT_REPO_ROOT/kaldi_amd/csrc/$l timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 2>&1 | grep "out+softmax\|out rows" | tail -5 || exit 1
done
