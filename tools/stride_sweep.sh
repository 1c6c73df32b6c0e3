#!/bin/bash
# stage 1 streaming rate against the rows per lane (= the stride between the lanes of a wave): N = C * 65536 rows,
# one round of 256 workgroups, d = 4 fp64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/stride_sweep.txt; : > $O
for c in 20 24 28 32 36 40 44 48 52 56 60 64 68 72 80 96 100 128 132 192 196 256 260; do
  CGPS_S1_C=$c python tools/time_mahal.py $((c*65536)) 2>/dev/null | awk -v c=$c '{ for(i=1;i<=NF;i++) if($i=="(min") { m=$(i+1); sub(/\)/,"",m); printf "C=%d rows=%d min %.1f us  stream GB/s (minus 26 us tail) %.0f\n", c, c*65536, m, c*65536*288/(m-26)/1e3 } }' >> $O
done
cat $O
