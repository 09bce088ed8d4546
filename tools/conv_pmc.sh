#!/bin/bash
# SQ / TCC counters of the conv kernels at the bench shape (tools/conv_bench.py), one rocprofv3 --pmc pass per set.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/conv_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/conv_bench.py > $OUT/p$i.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/conv_pmc/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv' not in n or 'layout' in n: continue
        agg.setdefault((n.replace('void (anonymous namespace)::', '').split('(')[0], r['Counter_Name']), []).append(float(r['Counter_Value']))
for (n, c), v in agg.items():
    print('%-42s %-30s %.5g  (%d)' % (n, c, sum(v) / len(v), len(v)))
P
