#!/bin/bash
# On the GPU box: counter passes over the calibration kernels (FETCH_SIZE and WRITE_SIZE in separate passes, plus the raw
# request counters), then the table.  Usage: tools/instr/calib/run_calib.sh [out file]
OUT=${1:-profiles/r06_counter_calibration.txt}
ROOT=$(pwd)
export TMPDIR=/tmp
python tools/instr/calib/traffic_calib.py build
k=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=$ROOT/gpurun_out/calib_$k; rm -rf $d; mkdir -p $d
  (cd /tmp && timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o c -- python $ROOT/tools/instr/calib/traffic_calib.py > $d.log 2>&1)
  echo "group $k [$grp] rc=$?"
  k=$((k+1))
done
python tools/instr/calib/summarise.py gpurun_out/calib_0 gpurun_out/calib_1 gpurun_out/calib_2 gpurun_out/calib_3 | tee $OUT
