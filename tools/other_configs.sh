#!/bin/bash
# The other workloads of BASELINE.json through the same bench.py (not headline lines): 1 / 2 / 4 views per GPU of the hand
# scene, the 500 k composite with 7 views, the 100 k object with one view, the close-up camera set with 8 / 1 views.  Usage: tools/other_configs.sh [ROUND]
RND=${1:-r06}
OUT=profiles/${RND}_other_configs
mkdir -p $OUT
run() { name=$1; shift; python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-hints-variant --trained-steps 0 "$@" > $OUT/$name.json 2>/dev/null
        python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-hints-variant --trained-steps 0 --profile-all "$@" 2>&1 >/dev/null | grep -v amdgpu.ids > $OUT/${name}_breakdown.txt
        python -c "import json;d=json.load(open('$OUT/$name.json'));print('$name',d['value'],d['ms_per_step'])"; }
run hand_v1 --views 1
run hand_v2 --views 2
run hand_v4 --views 4
run composite_500k_v7 --kind composite --gaussians 500000 --views 7
run object_100k_v1 --kind object --gaussians 100000 --views 1
# SURVEY 8(d)'s "close-up" camera set (0.45 m: the hand fills the frame, deep tile lists, a third of the rectangles > 64 tiles)
run closeup_v8 --cam-radius 0.45 --views 8
run closeup_v1 --cam-radius 0.45 --views 1
mkdir -p gpurun_out/$(basename $OUT); cp $OUT/* gpurun_out/$(basename $OUT)/
