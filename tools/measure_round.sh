#!/bin/bash
# Round measurement pass on the GPU box: tests, smoke, bench (with cpu baseline), rocprofv3 kernel
# stats of the same bench command, and the FETCH_SIZE / WRITE_SIZE counter passes.
# Usage: tools/measure_round.sh TAG      (outputs under gpurun_out/)
TAG=${1:-x}
RND=${2:-r01}
ROOT=$(pwd)
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/tests_$TAG.log
python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke_$TAG.log
tools/pmc_collect.sh traffic_$TAG "FETCH_SIZE" "WRITE_SIZE"
python tools/pmc_summary.py gpurun_out/pmc_traffic_${TAG}_0 gpurun_out/pmc_traffic_${TAG}_1 --json gpurun_out/pmc_traffic_$TAG.json --workload 300000,8,1920,1080 > gpurun_out/pmc_traffic_$TAG.txt
mkdir -p profiles; cp gpurun_out/pmc_traffic_$TAG.json profiles/r01_pmc_traffic.json   # bench reads this for roofline.traffic
rm -rf gpurun_out/prof_$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o run -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/prof_$TAG.log 2>&1)
python bench.py --steps 50 --warmup 5 --profile-all > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cat gpurun_out/tests_$TAG.log gpurun_out/smoke_$TAG.log | tail -5
tail -22 gpurun_out/bench_$TAG.err; cat gpurun_out/bench_$TAG.json
