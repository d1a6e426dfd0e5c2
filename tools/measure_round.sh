#!/bin/bash
# Round measurement pass on the GPU box: rocprofv3 kernel stats of the bench command, the PMC counter passes
# (separate passes, never combined with API traces), the bench itself (with the CPU oracle leg).
# Usage: tools/measure_round.sh TAG [ROUND]      (outputs under gpurun_out/, summaries copied to profiles/)
TAG=${1:-x}
RND=${2:-r06}
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p profiles gpurun_out
# counter passes: HBM traffic (two passes: TCC slots), instruction mix + activity
tools/pmc_collect.sh ${TAG} "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_0 gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3 \
    --json gpurun_out/pmc_${TAG}.json --workload 300000,8,1920,1080 > gpurun_out/pmc_${TAG}.txt
cp gpurun_out/pmc_${TAG}.json profiles/${RND}_pmc.json       # (on the box: so that the bench run below reads THIS pass; tools/pull_profiles.sh copies the summaries home)
cp gpurun_out/pmc_${TAG}.txt profiles/${RND}_${TAG}_pmc.txt
rm -rf gpurun_out/prof_$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o run -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hints-variant --trained-steps 0 > $ROOT/gpurun_out/prof_$TAG.log 2>&1)
cp $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1) profiles/${RND}_${TAG}_rocprofv3_kernel_stats.csv
python tools/instr/step_timeline.py $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1) > profiles/${RND}_${TAG}_step_timeline.txt
rm -f $(find gpurun_out/prof_$TAG -name "*kernel_trace.csv")
# the bench line as the driver runs it (events around the dominant kernel only), then the per-kernel breakdown
# (--profile-all: events around all ~26 launches of a step, which costs ~0.1 ms per step)
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err0
python bench.py --steps 50 --warmup 5 --profile-all --no-cpu-baseline --no-hints-variant > gpurun_out/bench_${TAG}_all.json 2> gpurun_out/bench_$TAG.err
cp gpurun_out/bench_$TAG.json profiles/${RND}_${TAG}_bench.json
grep -v amdgpu.ids gpurun_out/bench_$TAG.err > profiles/${RND}_${TAG}_bench_kernel_breakdown.txt
python tools/kernel_limits.py profiles/${RND}_pmc.json > profiles/${RND}_${TAG}_kernel_limits.txt
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${RND}_${TAG}_* profiles/${RND}_pmc.json gpurun_out/profiles_$TAG/
cat profiles/${RND}_${TAG}_kernel_limits.txt; cat profiles/${RND}_${TAG}_step_timeline.txt
tail -22 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json; head -30 gpurun_out/pmc_${TAG}.txt
