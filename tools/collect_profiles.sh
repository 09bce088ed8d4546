#!/bin/bash
# Round-3 rocprofv3 evidence (run on the GPU box from the repo root; outputs under gpurun_out/prof_r03/).
#   kernel-trace --stats of bench.py at the three shapes, and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the
#   stand-alone recurrence launches (tools/gru_step_timing.py: T=405, B=10, H=800).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_r03; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b10 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras > $OUT/b10.json 2> $OUT/b10.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras --batch-size 32 > $OUT/b32.json 2> $OUT/b32.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b64x15 -- python3 bench.py --steps 8 --warmup 0 --no-cpu-baseline --no-extras --batch-size 64 --fixed-seconds 15 > $OUT/b64x15.json 2> $OUT/b64x15.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 tools/gru_step_timing.py > $OUT/pmc_$c.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -path "*b*" -delete      # (large; the stats files are what is kept)
ls -R $OUT | head -50
