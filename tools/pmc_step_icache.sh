#!/bin/bash
# GPU box: instruction-cache and wait counters of the kernels inside the captured T30 L8 / T63 L16 steps
# (rocprofv3 --pmc of tools/dynamics_step_profile.py; counters only, no trace domains).   Usage: tools/pmc_step_icache.sh
export TMPDIR=/tmp
root=$PWD
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU\b" | sort -u | tr '\n' ' '; echo
for cfg in "t30 8" "t63 16"; do
  d=/tmp/pmcic_$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $d -o ic -- python $root/tools/dynamics_step_profile.py $cfg > $d.log 2>&1
  echo "== $cfg"; python $root/profiles/summarize_rocpd.py $(find $d -name "*_results.db") | grep "spdy::"
done
