cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_r04b; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b10 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras > $OUT/b10.json 2> $OUT/b10.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras --batch-size 32 > $OUT/b32.json 2> $OUT/b32.err
for c in FETCH_SIZE WRITE_SIZE; do
  BSZ=10 timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_b10 -- python3 tools/gru_step_timing.py > $OUT/pmc_${c}_b10.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -path "*/b*" -delete
BSZ=10 bash tools/gru_pmc.sh > $OUT/pmc_sq_gru_T405_B10.txt 2>&1
for w in fwd bwd; do echo "=== $w"; WHICH=$w python3 tools/gru_wave_timing.py 2>&1 | grep -v amdgpu.ids; done > $OUT/wave_timing_final.txt
ls -R $OUT | head -30
