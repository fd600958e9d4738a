#!/bin/bash
# rocprofv3 counter passes over the GMM scoring kernel (one pass per run, no tracing domains).
# usage: tools/pmc_passes.sh <tag> [perf_gmm args...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 tools/perf_gmm.py ${ARGS} > gpurun_out/pmc_${tag}_${name}.log 2>&1 || echo "pass $name failed"
}
ARGS="${*:-swbd_tri4 35554 3x8}"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq2 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT
run sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* > gpurun_out/pmc_${tag}_summary.txt 2>&1
cat gpurun_out/pmc_${tag}_summary.txt
