#!/usr/bin/env bash
# GPU box: same-box A/B of experiment builds (make -C speedy.f90_amd exp EXPNAME=x ...): headline rate and the two fused kernels'
# launch times for the product library and every build_dbg/libspdy_<name>.so named on the command line, two rounds.
#   bash tools/ab_libs.sh [--res t63] name1 name2 ...
res=t30; if [ "$1" = "--res" ]; then res=$2; shift 2; fi
for round in 1 2; do
  for n in base "$@"; do
    lib=speedy.f90_amd/libspdy.so; [ "$n" != base ] && lib=speedy.f90_amd/build_dbg/libspdy_$n.so
    SPDY_LIB=$PWD/$lib timeout 300 python bench.py --res $res --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms']
print('%-10s %.2f M rt/s  ' % ('$n', d['value']/1e6) + '  '.join('%s %.1f us' % (a, b*1e3) for a, b in sorted(k.items())))"
  done
done
