#!/bin/bash
# inverse_blocks: register-resident vs LDS-resident fused passes, per block size (A/B timing)
for c in "2 f64" "3 f64" "4 f64" "5 f64" "4 f32" "5 f32" "6 f32" "7 f32"; do
  set -- $c
  a=$(timeout -k 10 200 python tools/prof_case.py --op inverse_blocks --rows 1048576 --d $1 --dtype $2 --reps 20 | grep -o "[0-9.]* us")
  b=$(CGPS_INV_REG_MAX_BLOCK=0 timeout -k 10 200 python tools/prof_case.py --op inverse_blocks --rows 1048576 --d $1 --dtype $2 --reps 20 | grep -o "[0-9.]* us")
  echo "d=$1 $2 N=2^20: registers $a   LDS $b"
done
