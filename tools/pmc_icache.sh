cd /tmp; export TMPDIR=/tmp
for L in libspdy.so build_dbg/lib_v2.so; do
  tag=$(basename $L .so)
  SPDY_LIB=$GRAFT_REPO_ROOT/speedy.f90_amd/$L rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM -d /tmp/ic_$tag -o ic -- python $GRAFT_REPO_ROOT/bench.py --res t63 --no-cpu-baseline --steps 5 --warmup 2 > /tmp/ic_$tag.log 2>&1
  echo "== $L"; python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py $(find /tmp/ic_$tag -name "*_results.db") | grep -E "g2s_fused_t63" | head -12
done
