#!/bin/bash
# GPU box: tools/t63_exp.sh <tag> <variant>...  -- runs tools/t63_variants.py for speedy.f90_amd/build_dbg/libspdy_<variant>.so
# (experiment builds: make -C speedy.f90_amd exp EXPNAME=<variant> EXPFLAGS=-D...), twice each, interleaved, into gpurun_out/<tag>.txt
tag=$1; shift
out=gpurun_out/$tag.txt; : > $out
for rep in 1 2; do
  for v in "$@"; do
    SPDY_LIB=$PWD/speedy.f90_amd/build_dbg/libspdy_$v.so timeout 120 python tools/t63_variants.py ${T63_EXP_ARGS:-t63} 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
