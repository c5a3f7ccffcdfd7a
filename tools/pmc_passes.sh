#!/bin/bash
# tools/pmc_passes.sh -- rocprofv3 counter passes for the headline bench (one run per pass; PMC is
# never combined with tracing).  Run on the GPU box from the repo root; results in gpurun_out/pmc_*.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
CMD="python bench.py --steps 1 --warmup 0 --no-cpu --no-configs"
run() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- $CMD > gpurun_out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_$name.log; exit 1; }; echo "pass $name ok"; }
run sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE || exit 1
run sq2 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAIT_ANY SQ_INSTS_SALU || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
