cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/soak_r06b; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout 1500 python3 bench.py --no-extras --no-cpu-baseline --no-floor --no-f32-leg "$@" > $OUT/$name.json 2> $OUT/$name.err; python3 - $OUT/$name.json "$name" "$*" <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); c = r['config']
print('%-12s %-36s sync %9.1f frames/s %7.3f ms | deferred %9.1f %7.3f ms | fall-backs %s | loss %s'
      % (sys.argv[2], sys.argv[3], r['value'], r['ms_per_step'], c['deferred_readback']['frames_per_s'], c['deferred_readback']['ms_per_step'], c['persistent_to_step_fallbacks'], c['last_loss']))
PY
}
run b10_6000 --steps 6000
run b12_2000 --steps 2000 --batch-size 12
run b9_2000 --steps 2000 --batch-size 9
run b11_1000 --steps 1000 --batch-size 11
