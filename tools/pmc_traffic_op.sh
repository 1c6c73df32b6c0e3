#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one op through tools/prof_case.py
# usage (on the GPU box): bash tools/pmc_traffic_op.sh OUTDIR op [extra prof_case args]
set -e
OUT=$1; op=$2; shift 2; mkdir -p ${GRAFT_REPO_ROOT:-/root/repo}/$OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/${op}_$c -o p --output-format csv -- python3 $R/tools/prof_case.py --op $op --reps 3 "$@" > $R/$OUT/${op}_$c.log 2>&1
done
echo traffic passes done
