# Device assembly of one kernel: tools/kernel_asm.sh <mangled-name-substring> <out.s> [flags...]
# (SRC=<file under kaldi_amd/csrc> selects the source, default gmm_score.hip)
set -e
pat=$1; out=$2; shift 2
cd "$(dirname "$0")/../kaldi_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include "$@" --cuda-device-only -S -o /tmp/_k.s ${SRC:-gmm_score.hip} 2>/dev/null
awk -v pat="$pat" '$0 ~ "^_Z.*"pat".*:" {p=1} p {print} p && /^\.Lfunc_end/ {exit}' /tmp/_k.s > "$out"
grep -A30 "\.name: *_Z.*$pat" /tmp/_k.s | grep -E "vgpr_count|sgpr_count|spill|lds_size" | head -5
wc -l "$out"
