#!/bin/bash
# mahal_and_det at 2^20 rows for the given "d dtype" pairs: us per call (bench.py --rows path is d=4 only)
for c in "$@"; do
  set -- $c
  timeout -k 10 200 python tools/prof_case.py --op mahal_and_det --rows 1048576 --d $1 --dtype $2 --reps 50
done
