#!/bin/bash
# Run LOCALLY after a `gpurun ... tools/measure_round.sh TAG` call: gpurun merges only gpurun_out/ back, so the summaries
# that are to be judged are copied from there into profiles/ (tracked).   Usage: tools/pull_profiles.sh TAG [ROUND]
T=${1:?tag}; R=${2:-r02}
cp gpurun_out/pmc_$T.txt profiles/${R}_${T}_pmc.txt
cp gpurun_out/pmc_$T.json profiles/${R}_pmc.json
cp gpurun_out/bench_$T.json profiles/${R}_${T}_bench.json
grep -v amdgpu.ids gpurun_out/bench_$T.err > profiles/${R}_${T}_bench_kernel_breakdown.txt
f=$(find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${R}_${T}_rocprofv3_kernel_stats.csv
ls -la profiles | grep ${R}_${T}
