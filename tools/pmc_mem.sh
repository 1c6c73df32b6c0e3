#!/bin/bash
# Memory-system PMC passes (one rocprofv3 run per counter set) for one op of the hot path.
# usage (on the GPU box): bash tools/pmc_mem.sh OUTDIR op
set -e
OUT=$1; op=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY" \
           "TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN1 TCP_TA_TCP_STATE_READ" \
           "TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ" "TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_REQ TCC_BUSY" \
           "TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_64B TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/$OUT/${op}_m$i -o p --output-format csv -- python3 $R/tools/prof_case.py --op $op --reps 3 > $R/$OUT/${op}_m$i.log 2>&1 || echo "pass $i failed"
done
echo pmc mem done
