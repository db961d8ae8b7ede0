#!/bin/bash
# ablation timing of the fused direct kernel (debug; results are wrong by construction)
for fl in "" "-DABL_NOSTORE=1" "-DABL_NOLOAD=1" "-DABL_NOFFT=1" "-DABL_NOSTORE=1 -DABL_NOLOAD=1" "-DABL_NOSTORE=1 -DABL_NOLOAD=1 -DABL_NOFFT=1"; do
  rm -f speedy.f90_amd/build/spdy_kernels.o
  make -s -C speedy.f90_amd HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -w $fl" || exit 1
  echo "== flags: [$fl]"
  for i in 1 2; do timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['all_kernels_ms'])"; done
done
