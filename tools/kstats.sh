#!/bin/bash
# per-kernel durations of one op: tools/kstats.sh <op> [prof_case args]; summary -> gpurun_out/kstats_<op>.csv
R=${GRAFT_REPO_ROOT:-/root/repo}
op=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst_$op
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kst_$op -o p --output-format csv -- python3 $R/tools/prof_case.py --op $op --reps 20 "$@" > $R/gpurun_out/kstats_$op.log 2>&1
grep "us per call" $R/gpurun_out/kstats_$op.log
f=$(find /tmp/kst_$op -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/kst_$op -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then cp "$t" $R/gpurun_out/ktrace_$op.csv; fi
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/kstats_$op.csv; cut -d, -f1-6 "$f" | cut -c1-160; else echo "no kernel_stats.csv"; fi
