#!/bin/bash
# PMC counters of the round-5 3x3 probe (run on the GPU box): two passes of 8 SQ counters each.  usage: bash tools/band2pmc.sh [batch] (SHAPES=2 ...)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set -d /tmp/pm -o r -- python $R/tools/band2probe.py ${1:-64} > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pm -name "*.db" | head -1) 1
done
