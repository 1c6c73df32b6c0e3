#!/bin/bash
# A/B of stage 1's rows per lane for systems beyond one round of the chip (cgps_tile.h: long_chunk_*)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=gpurun_out/long_chunk_sweep.txt; : > $O
for n in $((1<<21)) $((1<<22)) $((1<<24)) 3000001; do
  CGPS_S1_LONG=0 python tools/time_mahal.py $n >> $O 2>&1
  python tools/time_mahal.py $n >> $O 2>&1
  CGPS_S1_TILES=512 python tools/time_mahal.py $n >> $O 2>&1
  CGPS_S1_NARROW=1 CGPS_S1_TILES=512 python tools/time_mahal.py $n >> $O 2>&1
  CGPS_S1_NARROW=1 CGPS_S1_TILES=1024 python tools/time_mahal.py $n >> $O 2>&1
done
python tools/time_mahal.py $((1<<21)) --shard >> $O 2>&1
CGPS_S1_LONG=0 python tools/time_mahal.py $((1<<21)) --shard >> $O 2>&1
for d in 2 3 5; do CGPS_S1_LONG=0 python tools/time_mahal.py $((1<<22)) $d >> $O 2>&1; python tools/time_mahal.py $((1<<22)) $d >> $O 2>&1; done
CGPS_S1_LONG=0 python tools/time_mahal.py $((1<<23)) 4 f32 >> $O 2>&1; python tools/time_mahal.py $((1<<23)) 4 f32 >> $O 2>&1
cat $O
