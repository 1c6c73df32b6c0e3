#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/pmc_leg; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace -d $R/$O/$c -o p --output-format csv -- python3 $R/tools/prof_leg_fused.py > $R/$O/$c.log 2>&1
done
cd $R; python tools/pmc_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
