#!/bin/bash
# GPU box: rocprofv3 evidence for the bench command (kernel trace + separate PMC passes), summarised into
# gpurun_out/<tag>.txt.  Usage: tools/profile_round.sh <tag> [bench args...]
# PMC passes never carry hip/hsa/sys trace options (kernel-trace only).
tag=${1:-prof}; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
# (--no-extras: only the metric's own launches, so that a kernel's average duration and counters are those of ONE batch size)
cmd="python $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-pmc $*"
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $OLDPWD/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras --no-pmc "$@" > $out/kt.log 2>&1
rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $out/fetch -o fetch -- $cmd > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $out/write -o write -- $cmd > $out/write.log 2>&1
rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES -d $out/sq -o sq -- $cmd > $out/sq.log 2>&1
rocprofv3 --kernel-trace --stats --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/sq2 -o sq2 -- $cmd > $out/sq2.log 2>&1
cd $OLDPWD
python profiles/summarize_rocpd.py $(find $out -name "*_results.db" | sort) > gpurun_out/$tag.txt
# the per-field lookup bench.py labels with `traffic_source` (copy to profiles/pmc_traffic.json together with the summary)
python tools/make_pmc_json.py gpurun_out/${tag}_pmc.json "profiles/$tag.txt (tools/profile_round.sh: rocprofv3 kernel-trace + PMC passes of bench.py $*)" profiles/pmc_traffic.json $(find $out -name "*_results.db" | sort) > /dev/null
grep -h "^{\"metric\"" $out/kt.log | tail -1 > gpurun_out/${tag}_bench.json   # (HIP-event times inside a profiled run are perturbed by the tool: only the kernel table above is evidence)
rm -rf $out/*/   # the raw databases are large; the summary is what gets committed
cat gpurun_out/$tag.txt | head -60
