#!/bin/bash
# GPU box: SQ / instruction-cache counters of the fused T63 kernels for one library build.
#   tools/t63_pmc.sh <tag> [variant]     (variant: speedy.f90_amd/build_dbg/libspdy_<variant>.so; default the product library)
# PMC passes carry --kernel-trace only.  Summary -> gpurun_out/<tag>.txt
tag=$1; v=$2
[ -n "$v" ] && export SPDY_LIB=$PWD/speedy.f90_amd/build_dbg/libspdy_$v.so
root=$PWD; out=$PWD/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $out/$n -o $n -- python $root/tools/t63_variants.py t63 1536 5 > $out/$n.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
run c SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC
run d SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
run e GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU2
cd $root
python profiles/summarize_rocpd.py $(find $out -name "*_results.db" | sort) | grep -E "fused_t63|^kernel" > gpurun_out/$tag.txt
rm -rf $out/*/
cat gpurun_out/$tag.txt
