#!/bin/bash
# SQ counters of the two persistent recurrence launches (tools/gru_step_timing.py: T = 405, H = 800, BSZ from the env),
# one rocprofv3 --pmc pass per counter set; prints per-kernel averages.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/gru_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/gru_step_timing.py > $OUT/p$i.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/gru_pmc/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gru_' not in n: continue
        k = 'fwd' if 'gru_fwd' in n else 'bwd'
        name = n.split('(')[1].split('::')[-1] if '(anonymous' in n else n
        agg.setdefault((k, n.replace('void (anonymous namespace)::', '').split('(')[0], r['Counter_Name']), []).append(float(r['Counter_Value']))
for (k, n, c), v in agg.items():
    print('%-45s %-28s %.5g  (%d launches)' % (n, c, sum(v) / len(v), len(v)))
P
