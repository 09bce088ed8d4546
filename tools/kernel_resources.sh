#!/bin/bash
# register / LDS / spill table of one csrc/*.hip file's kernels (device-only compile with -Rpass-analysis=kernel-resource-usage)
# usage: tools/kernel_resources.sh gru_persist.hip [name filter]
f=$1; pat=${2:-.}
out=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$(dirname "$0")/../include" -I "$(dirname "$0")/../aes-lac-2018_amd/csrc" \
  -I "$(dirname "$0")/../aes-lac-2018_amd/csrc/build" --cuda-device-only -c "$(dirname "$0")/../aes-lac-2018_amd/csrc/$f" -o $out/o.o \
  -Rpass-analysis=kernel-resource-usage $DS2_HIPCC_EXTRA 2> $out/res.txt
python3 - "$out/res.txt" "$pat" <<'PY'
import re, sys
cur = None
rows = {}
for line in open(sys.argv[1]):
    m = re.search(r'remark: \S+ +(Function Name|VGPRs|AGPRs|SGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)', line)
    if not m:
        continue
    if m.group(1) == 'Function Name':
        cur = m.group(2)
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1).split(' [')[0]] = m.group(2)
for k, v in rows.items():
    if re.search(sys.argv[2], k):
        print('%-90s %s' % (k[:90], ' '.join('%s=%s' % kv for kv in v.items())))
PY
rm -rf $out
