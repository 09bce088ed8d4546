#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r3h; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/b10 -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras > $OUT/b10.json 2> $OUT/b10.err
f=$(find $OUT/b10 -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py $f 3 --full > $OUT/timeline_full.txt
rm -f $f
head -45 $OUT/timeline_full.txt
