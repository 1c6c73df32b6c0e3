#!/bin/bash
# Memory-system PMC passes for one op at any size: bash tools/pmc_mem2.sh OUTDIR op [prof_case args]   (env passes through)
OUT=$1; op=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ" "TCC_EA0_RDREQ_32B TCC_TAG_STALL TCC_REQ TCC_BUSY" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace -d $R/$OUT/m$i -o p --output-format csv -- python3 $R/tools/prof_case.py --op $op --reps 3 "$@" > $R/$OUT/m$i.log 2>&1 || echo "pass $i failed"
done
