#!/bin/bash
# A/B of library variants on one box, one call: per variant the bench's step time and the kernels named by the pattern.
# Usage (GPU box): tools/instr/ab.sh "variant1 variant2" "k_bin|k_dbin" [extra bench flags]
PAT=${2:-k_blend}
for v in "" $1; do
  MANUS_HIP_VARIANT=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-hints-variant --profile-all ${@:3} 2>/tmp/ab.err >/tmp/ab.json
  line=$(grep -E "^($PAT)" /tmp/ab.err | awk '{printf "%s %s | ", $1, $7}')
  plain=$(MANUS_HIP_VARIANT=$v python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-hints-variant ${@:3} 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'])")
  echo "variant=${v:-product} step: $plain | $line"
done
