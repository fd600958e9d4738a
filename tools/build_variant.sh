# Builds kaldi_amd/csrc/libamlike_<tag>.so: the library with gmm_score.hip (or the file named by
# VARIANT_SRC) recompiled with extra flags, for A/B timing with AMLIKE_LIB_PATH (tools/ab.sh).
#   tools/build_variant.sh <tag> <flags...>
set -e
tag=$1; shift
cd "$(dirname "$0")/../kaldi_amd/csrc"
src=${VARIANT_SRC:-gmm_score.hip}
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include "$@" -c "$src" -o "build/variants/${src}.${tag}.o"
objs=""
for o in build/*.o; do
  case "$o" in build/${src}.o) objs="$objs build/variants/${src}.${tag}.o";; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "libamlike_${tag}.so" $objs
echo "built kaldi_amd/csrc/libamlike_${tag}.so"
