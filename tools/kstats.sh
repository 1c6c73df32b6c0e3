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
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/kstats_$op.csv; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'cgps' in r['Name']:
        print('%-70s calls %4s avg %9.2f us min %9.2f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
else echo "no kernel_stats.csv"; fi
