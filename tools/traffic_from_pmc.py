#!/usr/bin/env python
"""profiles/rNN_traffic*.json from the two rocprofv3 PMC passes of tools/gru_step_timing.py.

    python tools/traffic_from_pmc.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <previous.json> [note]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (MI355X_MICROARCH.md: both counters are in KiB and FETCH_SIZE counts
half of the wide reads on gfx950).  The shape, the source line and the ALGORITHMIC bytes per launch are
carried over from the previous round's file: they belong to the shape, not to the kernel's implementation.
"""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and 'persistent' in r['Kernel_Name']:
            name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            acc[name].append(float(r['Counter_Value']))
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    prev = json.load(open(sys.argv[3]))
    out = {'shape': prev['shape'], 'source': prev['source']}
    for name in sorted(fetch):
        key = name.split('<')[0]
        fs, ws = fetch[name], write.get(name, [])
        n = min(len(fs), len(ws))
        if n == 0:
            continue
        fk, wk = sum(fs[-n:]) / n, sum(ws[-n:]) / n
        out[key] = {'instantiation': name, 'FETCH_SIZE_KiB': fk, 'WRITE_SIZE_KiB': wk, 'launches': n,
                    'traffic_bytes_per_launch': int((2 * fk + wk) * 1024)}
        if key in prev and 'algorithmic_hbm_bytes_per_launch' in prev[key]:
            out[key]['algorithmic_hbm_bytes_per_launch'] = prev[key]['algorithmic_hbm_bytes_per_launch']
    if len(sys.argv) > 4:
        out['_note'] = sys.argv[4]
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
