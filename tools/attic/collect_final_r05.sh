# Round-5 final runs (GPU box, from the repo root): the default bench line, per-wave stamps, the soak.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/final_r05; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
WHICH=fwd python3 tools/gru_wave_timing.py > $OUT/wave_timing_fwd.txt 2>&1
WHICH=bwd python3 tools/gru_wave_timing.py > $OUT/wave_timing_bwd.txt 2>&1
soak() { # name, args
  python3 bench.py --no-extras --no-cpu-baseline "${@:2}" > $OUT/soak_$1.json 2> $OUT/soak_$1.err
  python3 - "$1" "$OUT/soak_$1.json" <<'P'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
c = d['config']
print('%-14s steps %5d x 3 legs  %9.1f frames/s  %8.3f ms/step (sync per step)  deferred %9.1f  last loss %.4f  fall-backs %d'
      % (sys.argv[1], d['steps'], d['value'], d['ms_per_step'], c['deferred_readback']['frames_per_s'], c['last_loss'], c['persistent_to_step_fallbacks']))
P
}
{
soak b10 --steps 1500
soak b12 --steps 500 --batch-size 12
soak b9 --steps 300 --batch-size 9
soak b8x15 --steps 200 --batch-size 8 --fixed-seconds 15
soak b32 --steps 200 --batch-size 32
soak b64x15 --steps 40 --batch-size 64 --fixed-seconds 15
} > $OUT/soak.txt 2>&1
cat $OUT/soak.txt
tail -c 400 $OUT/bench_default.json
