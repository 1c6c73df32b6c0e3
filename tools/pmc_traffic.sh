#!/bin/bash
# HBM traffic of the headline command: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; on gfx950 read bytes = 2 x FETCH_SIZE).
# usage (on the GPU box): bash tools/pmc_traffic.sh OUTDIR
set -e
OUT=$1; mkdir -p ${GRAFT_REPO_ROOT:-/root/repo}/$OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/$OUT/$c -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/$OUT/$c.log 2>&1
done
echo traffic passes done
