#!/bin/bash
# stage 1 with long chunks: skew of the lanes' chunk starts (CGPS_S1_SKEW rows, CGPS_S1_SKEWP period), d = 4 fp64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/skew_sweep.txt; : > $O
run() { env "$@" python tools/time_mahal.py $N 2>/dev/null | sed 's/rel err.*//' >> $O; }
N=$((1<<21))
CGPS_S1_LONG=0 python tools/time_mahal.py $N 2>/dev/null | sed 's/rel err.*//' >> $O
for cfg in "0 8" "4 2" "4 4" "4 8" "8 4" "8 2" "16 2" "12 2"; do set -- $cfg; run CGPS_S1_SKEW=$1 CGPS_S1_SKEWP=$2; done
N=$((1<<22))
CGPS_S1_LONG=0 python tools/time_mahal.py $N 2>/dev/null | sed 's/rel err.*//' >> $O
for cfg in "0 8" "4 8" "8 8" "4 16" "8 4" "16 4" "32 2"; do set -- $cfg; run CGPS_S1_SKEW=$1 CGPS_S1_SKEWP=$2; done
N=$((1<<24))
CGPS_S1_LONG=0 python tools/time_mahal.py $N 2>/dev/null | sed 's/rel err.*//' >> $O
for cfg in "0 8" "4 8" "16 16" "32 8" "4 64"; do set -- $cfg; run CGPS_S1_SKEW=$1 CGPS_S1_SKEWP=$2; done
N=3000001
for cfg in "0 8" "4 8"; do set -- $cfg; run CGPS_S1_SKEW=$1 CGPS_S1_SKEWP=$2; done
cat $O
