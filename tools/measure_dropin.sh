#!/bin/bash
# The zero-change drop-in route under the profiler: bench line + per-kernel breakdown + rocprofv3 kernel stats + one step's timeline
# (kernel trace between two k_skin_fwd24x8 launches: library kernels, torch glue kernels and the gaps the host leaves).
# Usage: tools/measure_dropin.sh TAG [ROUND]
TAG=${1:-a}; RND=${2:-r06}
ROOT=$(pwd); export TMPDIR=/tmp
OUT=profiles/${RND}_other_configs; mkdir -p $OUT gpurun_out
python bench.py --route dropin --steps 40 --warmup 5 --profile-all > $OUT/dropin_$TAG.json 2> gpurun_out/dropin_$TAG.err
grep -v "amdgpu.ids\|UserWarning\|Consider using\|finite_grads" gpurun_out/dropin_$TAG.err > $OUT/dropin_${TAG}_breakdown.txt
rm -rf gpurun_out/prof_dropin_$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_dropin_$TAG -o run -- python $ROOT/bench.py --route dropin --steps 20 --warmup 5 > $ROOT/gpurun_out/prof_dropin_$TAG.log 2>&1)
cp $(find gpurun_out/prof_dropin_$TAG -name "*kernel_stats.csv" | head -1) $OUT/dropin_${TAG}_rocprofv3_kernel_stats.csv
python tools/instr/step_timeline.py $(find gpurun_out/prof_dropin_$TAG -name "*kernel_trace.csv" | head -1) -20 > $OUT/dropin_${TAG}_step_timeline.txt
rm -f $(find gpurun_out/prof_dropin_$TAG -name "*kernel_trace.csv")
mkdir -p gpurun_out/dropin_$TAG; cp $OUT/dropin_${TAG}* gpurun_out/dropin_$TAG/
cat $OUT/dropin_${TAG}_step_timeline.txt; cat $OUT/dropin_${TAG}_breakdown.txt; cat $OUT/dropin_$TAG.json
