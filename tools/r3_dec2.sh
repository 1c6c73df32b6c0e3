#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for d in 7 8; do for n in 1024 16384 65536 262144; do
  python tools/prof_case.py --op decompose --d $d --rows $n --reps 20 2>/dev/null
  CGPS_LEVELWISE_SOLVE=1 python tools/prof_case.py --op decompose --d $d --rows $n --reps 20 2>/dev/null
done; done
