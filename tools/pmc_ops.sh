#!/bin/bash
# PMC passes (one rocprofv3 run per counter set and op) for the ops of the hot path.
# usage (on the GPU box): bash tools/pmc_ops.sh OUTDIR op1 op2 ...
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for op in "$@"; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS \
    --kernel-trace -d $R/$OUT/${op}_sq1 -o p --output-format csv -- python3 $R/tools/prof_case.py --op $op --reps 3 > $R/$OUT/${op}_sq1.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES \
    --kernel-trace -d $R/$OUT/${op}_sq2 -o p --output-format csv -- python3 $R/tools/prof_case.py --op $op --reps 3 > $R/$OUT/${op}_sq2.log 2>&1
done
echo pmc done
