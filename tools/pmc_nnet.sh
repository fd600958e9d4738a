#!/bin/bash
# rocprofv3 counter passes over the nnet2 forward kernels (one pass per run, no tracing domains).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-nnet}
run() { name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_${name} -- python3 bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > gpurun_out/pmc_${tag}_${name}.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
run grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* > gpurun_out/pmc_${tag}_summary.txt 2>&1
grep -A20 "gemm" gpurun_out/pmc_${tag}_summary.txt | head -24
