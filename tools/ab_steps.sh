#!/usr/bin/env bash
# GPU box: captured step times (T63 L16 / L8, T30 L8) for the product library and experiment builds, two rounds.
#   bash tools/ab_steps.sh name1 name2 ...      (build_dbg/libspdy_<name>.so from `make -C speedy.f90_amd exp EXPNAME=<name> ...`)
for round in 1 2; do
  for n in base "$@"; do
    lib=; [ "$n" != base ] && lib=$PWD/speedy.f90_amd/build_dbg/libspdy_$n.so
    a=$(SPDY_LIB=$lib python tools/t63_steps.py 2>/dev/null | grep step_ | awk '{printf "%s %s  ", $1, $2}')
    b=$(SPDY_LIB=$lib python tools/t30_small_batch.py 2>/dev/null | grep step_t30 | awk '{print "step_t30_l8", $2}')
    echo "$n: $a $b"
  done
done
