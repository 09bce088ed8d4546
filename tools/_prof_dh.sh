cd $GRAFT_REPO_ROOT
echo "=== ablations (faultinject build): DBG 0 / 2 (skeleton) / 2048 (no validation) / 4096 (no MFMA); SPARE 0 and 82"
for sp in 0 82; do for dh in 0 1; do for dbg in 0 2 2048 4096; do
  echo -n "dh=$dh "; DS2_GRU_BWD_DH=$dh DS2_LIB_VARIANT=faultinject DS2_GRU_DBG=$dbg SPARE_CUS=$sp BSZ=10 TSTEPS=405 timeout 120 python tools/gru_step_timing.py 2>&1 | tail -1
done; done; done
echo "=== wave stamps dgh"; DS2_GRU_BWD_DH=0 WHICH=bwd timeout 120 python tools/gru_wave_timing.py 2>&1 | tail -28
echo "=== wave stamps dh"; DS2_GRU_BWD_DH=1 WHICH=bwd timeout 120 python tools/gru_wave_timing.py 2>&1 | tail -28
