#!/bin/bash
# k_blend_bwd knock-outs (WRONG RESULTS, cost bounds only): per variant the kernel's mean duration by HIP events in the bench loop.
# Build first: for k in 1 3 4 8 16 32; do tools/instr/build_variant.sh ko$k raster_bwd "-DBWD_KO=$k"; done
# Usage (GPU box): tools/instr/ko_bwd.sh "ko1 ko3 ..." > gpurun_out/ko_bwd.txt
for v in "" $1; do
  line=$(MANUS_HIP_VARIANT=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-hints-variant --profile-all 2>&1 >/dev/null | grep -E "^(k_blend_bwd|k_blend_fwd|k_inst_gather|library kernels)" | tr '\n' '|')
  echo "variant=${v:-product} $line"
done
