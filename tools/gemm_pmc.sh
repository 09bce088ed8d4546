#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/gemm_pmc; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/gemm_pmc.py > $OUT/p$i.log 2>&1
done
python3 - <<'P'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/gemm_pmc/p*/**/*counter_collection.csv', recursive=True)):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if 'gemm' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'][:70], r['Grid_Size'], r['Counter_Name'])
        agg.setdefault(key, []).append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f.split('/')[2], k[0], k[1], k[2], '%.4g' % (sum(v) / len(v)), len(v))
P
