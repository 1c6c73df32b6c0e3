#!/bin/bash
# mahal_and_det across system sizes on one GPU (bench.py --rows): us per call and fraction of 8 TB/s
# usage: bash tools/size_sweep.sh [log2 sizes ...]
sizes="$@"
[ -z "$sizes" ] && sizes="14 16 18 19 20 21 22 23 24"
for lg in $sizes; do
  python bench.py --rows $((1<<lg)) --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('N=2^$lg  %9.1f us  %7.1f GB/s  whole-op frac %.3f  kernel frac %.3f' % (j['ms_per_step']*1e3, j['value'], j['roofline_frac_whole_op'], j['roofline']['frac']))"
done
