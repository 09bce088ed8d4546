# Round-6 soak + survey-protocol line (run on the GPU box from the repo root; outputs under gpurun_out/final_r06/)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/final_r06; rm -rf $OUT; mkdir -p $OUT
timeout 600 python3 bench.py --protocol survey --no-cpu-baseline --no-extras > $OUT/survey_protocol.json 2> $OUT/survey.err
run() { name=$1; shift; timeout 900 python3 bench.py --no-extras --no-cpu-baseline --no-floor "$@" > $OUT/$name.json 2> $OUT/$name.err; python3 - $OUT/$name.json "$name" "$*" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
c = r['config']
print('%-14s %-44s sync %9.1f frames/s %7.3f ms | deferred %9.1f %7.3f ms | f32-GEMM leg %s | fall-backs %s | loss %s'
      % (sys.argv[2], sys.argv[3], r['value'], r['ms_per_step'], c['deferred_readback']['frames_per_s'], c['deferred_readback']['ms_per_step'],
         (c['gemm_arithmetic']['same_steps_on_f32_input_mfma_gemms'] or {}).get('ms_per_step'), c['persistent_to_step_fallbacks'], c['last_loss']))
PY
}
run b10_1500 --steps 1500
run b12_500 --steps 500 --batch-size 12
run b9_300 --steps 300 --batch-size 9
run b8x15_200 --steps 200 --batch-size 8 --fixed-seconds 15
run b32_200 --steps 200 --batch-size 32
run b64x15_40 --steps 40 --batch-size 64 --fixed-seconds 15
