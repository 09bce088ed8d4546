# Round-5 evidence (run on the GPU box from the repo root; outputs under gpurun_out/prof_r05/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_r05; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b10 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras > $OUT/b10.json 2> $OUT/b10.err
python3 tools/step_idle_gaps.py $OUT/b10/*/*kernel_trace.csv > $OUT/b10_idle_gaps.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32 -- python3 bench.py --steps 24 --warmup 0 --no-cpu-baseline --no-extras --batch-size 32 > $OUT/b32.json 2> $OUT/b32.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b64x15 -- python3 bench.py --steps 8 --warmup 0 --no-cpu-baseline --no-extras --batch-size 64 --fixed-seconds 15 > $OUT/b64x15.json 2> $OUT/b64x15.err
# HBM-side traffic of the recurrence launches: the backward launch in the form the lower layers take (174 workgroups) and in the
# top layer's (240), separate PMC passes per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass)
for sp in 82 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    BSZ=10 SPARE_CUS=$sp timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_b10_spare$sp -- python3 tools/gru_step_timing.py > $OUT/pmc_${c}_b10_spare$sp.log 2>&1
  done
done
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
for sp in 82 52 0; do BSZ=10 SPARE_CUS=$sp python3 tools/gru_step_timing.py 2>&1 | tail -1; done > $OUT/step_timing_forms.txt
for b in 4 8 9 12; do BSZ=$b python3 tools/gru_step_timing.py 2>&1 | tail -1; done >> $OUT/step_timing_forms.txt
BSZ=10 bash tools/gru_pmc.sh > $OUT/pmc_sq_gru_T405_B10.txt 2>&1
ls -R $OUT | head -60
