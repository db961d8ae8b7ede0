#!/bin/bash
# GPU box: per-kernel durations inside the captured dynamical-core step (rocprofv3 --kernel-trace --stats of
# tools/dynamics_step_profile.py) at T63 L16, T63 L8, T30 L8, T30 L16 -> gpurun_out/<tag>.txt.   Usage: tools/profile_step.sh <tag>
tag=${1:-step}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
root=$PWD
echo "Captured dynamical-core step (tools/dynamics_step_profile.py <res> <kx> under rocprofv3 --kernel-trace --stats): kernel, calls, total us, avg us, % of the run" > gpurun_out/$tag.txt
cd /tmp
for cfg in "t63 16" "t63 8" "t30 8" "t30 16"; do
    d=$out/$(echo $cfg | tr ' ' '_')
    rocprofv3 --kernel-trace --stats -d $d -o kt -- python $root/tools/dynamics_step_profile.py $cfg > $d.log 2>&1
    echo "== $cfg   $(grep -h us_per_step $d.log | tail -1)" >> $root/gpurun_out/$tag.txt
    python $root/profiles/summarize_rocpd.py $(find $d -name "*_results.db") | grep "spdy::" >> $root/gpurun_out/$tag.txt
done
cd $root
rm -rf $out
cat gpurun_out/$tag.txt
