#!/bin/bash
# Round-4 rocprofv3 evidence (run on the GPU box from the repo root; outputs under gpurun_out/prof_r04/).
#   kernel-trace --stats of bench.py at the three shapes, two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the stand-alone
#   recurrence launches (tools/gru_step_timing.py: T=405, H=800) at B=10 and B=32, and the SQ counter passes at B=32.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_r04; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b10 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras > $OUT/b10.json 2> $OUT/b10.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras --batch-size 32 > $OUT/b32.json 2> $OUT/b32.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b64x15 -- python3 bench.py --steps 8 --warmup 0 --no-cpu-baseline --no-extras --batch-size 64 --fixed-seconds 15 > $OUT/b64x15.json 2> $OUT/b64x15.err
for b in 10 32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    BSZ=$b timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_b$b -- python3 tools/gru_step_timing.py > $OUT/pmc_${c}_b$b.log 2>&1
  done
done
find $OUT -name "*kernel_trace.csv" -path "*/b*" -delete      # (large; the stats files are what is kept)
BSZ=32 bash tools/gru_pmc.sh > $OUT/pmc_sq_gru_T405_B32.txt 2>&1
BSZ=10 bash tools/gru_pmc.sh > $OUT/pmc_sq_gru_T405_B10.txt 2>&1
for t in 100 405 810; do TSTEPS=$t python3 tools/gru_step_timing.py 2>&1 | tail -1; done > $OUT/launch_overhead.txt
ls -R $OUT | head -60
