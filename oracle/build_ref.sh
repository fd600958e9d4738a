#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds the real reference library (the subset of
# Kaldi sources the acoustic-likelihood path needs) from the sources where they
# lie, plus oracle/ref_driver.cc, into oracle/_ref/.  Nothing under oracle/ is
# used by the product path; only tests/, smoke() and bench.py's cpu_baseline
# leg run the resulting binary.
#
#   oracle/build_ref.sh [/root/reference]
#
# Flags follow the reference's src/makefiles/linux_clapack.mk (g++ -O2, CLAPACK
# headers shipped in tools/CLAPACK, HAVE_CUDA undefined so CuMatrix takes its
# CPU branches).  BLAS/LAPACK = MKL's single dynamic library from the image's
# conda (run with MKL_THREADING_LAYER=SEQUENTIAL).
#
# Excluded on purpose: every translation unit that includes
# hmm/transition-model.h (it needs OpenFst's fst/fst-decl.h, which the image
# lacks): hmm/*, gmm/decodable-am-diag-gmm.h, nnet2/nnet-compute.cc, tree/tree-renderer.cc.
set -euo pipefail
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
MKL=/opt/conda/lib/libmkl_rt.so
if [ ! -d "$REF/src" ]; then echo "build_ref: no reference at $REF; keeping prebuilt $OUT" >&2; exit 0; fi
if [ ! -f "$MKL" ]; then echo "build_ref: $MKL missing; reference build not possible" >&2; exit 3; fi
mkdir -p "$OUT/obj" "$OUT/mkl"

CXXFLAGS="-std=gnu++11 -O2 -msse -msse2 -pthread -I$REF/src -I$REF/tools/CLAPACK \
 -DKALDI_DOUBLEPRECISION=0 -DHAVE_POSIX_MEMALIGN -DHAVE_EXECINFO_H=1 -DHAVE_CXXABI_H -DHAVE_CLAPACK -w"

srcs=()
for d in base matrix util thread tree gmm cudamatrix; do
  for f in "$REF/src/$d"/*.cc; do
    b="$(basename "$f")"
    case "$b" in *-test.cc|*-speed-test.cc|tree-renderer.cc|decodable-am-diag-gmm.cc) continue;; esac
    srcs+=("$f")
  done
done
for b in nnet-component nnet-nnet am-nnet nnet-precondition nnet-precondition-online; do
  srcs+=("$REF/src/nnet2/$b.cc")
done

# SpliceFrames (front end of the GMM systems) and the one function of feat/ it links against
for b in feature-functions mel-computations; do
  srcs+=("$REF/src/feat/$b.cc")
done

compile_one() {
  f="$1"; o="$OUT/obj/$(basename "$(dirname "$f")")_$(basename "${f%.cc}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then g++ $CXXFLAGS -c "$f" -o "$o"; fi
}
export -f compile_one; export OUT CXXFLAGS
printf '%s\n' "${srcs[@]}" | xargs -P "${JOBS:-8}" -I{} bash -c 'compile_one "$@"' _ {}

# MKL: a private directory of symlinks, so that the conda libstdc++ next to it
# is never picked up.
for f in /opt/conda/lib/libmkl_*.so* /opt/conda/lib/libiomp5.so; do
  [ -e "$f" ] && ln -sf "$f" "$OUT/mkl/$(basename "$f")"
done

g++ $CXXFLAGS "$HERE/ref_driver.cc" "$OUT"/obj/*.o "$MKL" -lm -lpthread -ldl \
  -Wl,-rpath,"$OUT/mkl" -Wl,-rpath,'$ORIGIN/mkl' -o "$OUT/ref_driver"
echo "built $OUT/ref_driver"
