#!/bin/bash
# The training step WITH the optimizer under the profiler (verdict r04 item 1c): rocprofv3 kernel stats + one step's
# timeline of `bench.py --optimizer`, the per-kernel HIP-event breakdown, and the same box's plain lines beside it.
# Usage: tools/measure_optimizer.sh TAG [ROUND] [extra bench flags]   (outputs under gpurun_out/ and profiles/)
TAG=${1:-x}
RND=${2:-r06}
shift; shift
EXTRA="$@"
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p profiles gpurun_out
rm -rf gpurun_out/prof_opt_$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_opt_$TAG -o run -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --optimizer $EXTRA > $ROOT/gpurun_out/prof_opt_$TAG.log 2>&1)
cp $(find gpurun_out/prof_opt_$TAG -name "*kernel_stats.csv" | head -1) profiles/${RND}_${TAG}_optimizer_rocprofv3_kernel_stats.csv
python tools/instr/step_timeline.py $(find gpurun_out/prof_opt_$TAG -name "*kernel_trace.csv" | head -1) > profiles/${RND}_${TAG}_optimizer_step_timeline.txt
rm -f $(find gpurun_out/prof_opt_$TAG -name "*kernel_trace.csv")
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --optimizer $EXTRA > profiles/${RND}_${TAG}_optimizer_bench.json 2> gpurun_out/opt_$TAG.err0
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --optimizer --profile-all $EXTRA 2>&1 >/dev/null | grep -v amdgpu.ids > profiles/${RND}_${TAG}_optimizer_kernel_breakdown.txt
python bench.py --steps 50 --warmup 5 --no-cpu-baseline $EXTRA > profiles/${RND}_${TAG}_optimizer_ref_noopt.json 2>/dev/null
cp profiles/${RND}_${TAG}_optimizer_* gpurun_out/
cat profiles/${RND}_${TAG}_optimizer_step_timeline.txt
cat profiles/${RND}_${TAG}_optimizer_kernel_breakdown.txt
python - <<PY
import json
for f in ("optimizer_bench", "optimizer_ref_noopt"):
    d = json.load(open("profiles/${RND}_${TAG}_%s.json" % f)); print(f, d["value"], d["ms_per_step"])
PY
