# Round-6 evidence (run on the GPU box from the repo root; outputs under gpurun_out/prof_r06/).  The bench legs that launch
# ABLATED kernels (the floor leg) or another kernel family (the f32-GEMM leg) under the product kernels' names are switched
# off, so the kernel-stats CSVs contain product launches only.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_r06; rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-extras --no-floor --no-f32-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b10 -- python3 bench.py --steps 24 --warmup 0 $B > $OUT/b10.json 2> $OUT/b10.err
python3 tools/step_idle_gaps.py $OUT/b10/*/*kernel_trace.csv > $OUT/b10_idle_gaps.txt 2>&1
if [ "$1" != "quick" ]; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b32 -- python3 bench.py --steps 24 --warmup 0 $B --batch-size 32 > $OUT/b32.json 2> $OUT/b32.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b64x15 -- python3 bench.py --steps 8 --warmup 0 $B --batch-size 64 --fixed-seconds 15 > $OUT/b64x15.json 2> $OUT/b64x15.err
for sp in 82 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    BSZ=10 SPARE_CUS=$sp timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_b10_spare$sp -- python3 tools/gru_step_timing.py > $OUT/pmc_${c}_b10_spare$sp.log 2>&1
  done
done
for sp in 82 52 0; do BSZ=10 SPARE_CUS=$sp python3 tools/gru_step_timing.py 2>&1 | tail -1; done > $OUT/step_timing_forms.txt
for b in 4 8 9 12; do BSZ=$b python3 tools/gru_step_timing.py 2>&1 | tail -1; done >> $OUT/step_timing_forms.txt
fi
# keep ONE trace (the b10 run's) for tools/step_idle_gaps.py readers, compressed; drop the rest
gzip -c $OUT/b10/*/*kernel_trace.csv > $OUT/b10_kernel_trace.csv.gz
find $OUT -name "*kernel_trace.csv" -not -path "*pmc*" -delete
find $OUT -name "*.db" -delete
ls -R $OUT | head -60
