#!/bin/bash
# Collect per-kernel PMC counters for the bench workload, one rocprofv3 pass per counter group
# (counters are never combined with API traces).  Usage: tools/pmc_collect.sh TAG "C1 C2" "C3" ...
# Output: gpurun_out/pmc_<TAG>_<k>/ ; summarise with tools/pmc_summary.py
set -u
TAG=$1; shift
ROOT=$(pwd)
export TMPDIR=/tmp
k=0
for grp in "$@"; do
  out=$ROOT/gpurun_out/pmc_${TAG}_$k
  rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o runc -- \
      python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-hints-variant --trained-steps 0 ${PMC_BENCH_FLAGS:-} > $out.log 2>&1)
  echo "group $k [$grp] rc=$?"
  k=$((k+1))
done
