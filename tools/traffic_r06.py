#!/usr/bin/env python
"""profiles/r06_traffic.json from the four rocprofv3 PMC passes of tools/collect_profiles_r06.sh (FETCH_SIZE / WRITE_SIZE, each
with the backward launch in the lower layers' form, SPARE_CUS=82, and in the top layer's, SPARE_CUS=0).

    python tools/traffic_r06.py gpurun_out/prof_r06 > profiles/r06_traffic.json

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch (MI355X_MICROARCH.md: both counters are in KiB and FETCH_SIZE counts half
of the wide reads on gfx950)."""
import csv, glob, json, os, sys
from collections import defaultdict

T, B, H = 405, 10, 800
PLANE = T * B * 2 * H * 4            # bytes of one (T, B, 2, H) fp32 plane


def per_kernel(dirname, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == counter and 'persistent' in r['Kernel_Name']:
                name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
                acc[name].append(float(r['Counter_Value']))
    return acc


def entry(root, spare, want):
    f = per_kernel(os.path.join(root, 'pmc_FETCH_SIZE_b10_spare%d' % spare), 'FETCH_SIZE')
    w = per_kernel(os.path.join(root, 'pmc_WRITE_SIZE_b10_spare%d' % spare), 'WRITE_SIZE')
    for name in f:
        if name.startswith(want) and name in w:
            n = min(len(f[name]), len(w[name]))
            fk, wk = sum(f[name][-n:]) / n, sum(w[name][-n:]) / n
            return {'instantiation': name, 'FETCH_SIZE_KiB': fk, 'WRITE_SIZE_KiB': wk, 'launches': n,
                    'traffic_bytes_per_launch': int((2 * fk + wk) * 1024)}
    raise SystemExit('no %s in %s (spare %d)' % (want, root, spare))


def main():
    root = sys.argv[1]
    below = entry(root, 82, 'gru_bwd_persistent6_kernel')
    top = entry(root, 0, 'gru_bwd_persistent6_kernel')
    fwd = entry(root, 82, 'gru_fwd_persistent5_kernel')
    # what crosses the memory side at least once: 6 planes of saved activations loaded (r, z, n, gh_n, h_prev, d_out), 4 stored
    # (d(gi) x 3, d(gh_n)) -- minus nothing -- and, for this form, the three coefficient planes loaded
    alg_bwd = int(9.5 * PLANE)                       # as rounds 1-5 quoted it (d_out is (T, B, H): half a plane)
    alg_bwd_dh = alg_bwd + 3 * PLANE
    # an exchange through memory between CUs on EIGHT XCDs with non-coherent L2s: the payload (ONE plane now) into every XCD,
    # written once as payload and once as canary; the coefficient planes (3) into every XCD's L2; saved activations as above
    floor_dh = int((8 * 1 + 1 + 1 + 8 * 3 + 10) * PLANE)
    mean = {k: (top[k] + 4 * below[k]) / 5 for k in ('FETCH_SIZE_KiB', 'WRITE_SIZE_KiB')}
    out = {
        'shape': {'T': T, 'B': B, 'H': H},
        'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/gru_step_timing.py (BSZ=10, '
                  'SPARE_CUS=82 and 0; tools/collect_profiles_r06.sh); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024',
        'backward_recurrence_launch': {
            'instantiation': 'mean launch of a step: 1 x %s + 4 x %s' % (top['instantiation'], below['instantiation']),
            'FETCH_SIZE_KiB': mean['FETCH_SIZE_KiB'], 'WRITE_SIZE_KiB': mean['WRITE_SIZE_KiB'], 'launches': below['launches'],
            'traffic_bytes_per_launch': int((2 * mean['FETCH_SIZE_KiB'] + mean['WRITE_SIZE_KiB']) * 1024),
            'algorithmic_hbm_bytes_per_launch': alg_bwd,
            'algorithmic_hbm_bytes_per_launch_with_coefficient_planes': alg_bwd_dh,
            'xcd_replicated_floor_bytes_per_launch': floor_dh},
        'gru_bwd_persistent6_kernel': dict(below, algorithmic_hbm_bytes_per_launch=alg_bwd),
        'gru_bwd_persistent6_kernel_top_layer': dict(top, algorithmic_hbm_bytes_per_launch=alg_bwd),
        'gru_fwd_persistent5_kernel': dict(fwd, algorithmic_hbm_bytes_per_launch=int((3 + 5 + 3) * PLANE - 0.0),
                                           note='3 planes of gi loaded; r, z, n, gh_n, h stored; this round also the three '
                                                'coefficient planes (a training pass): 11 planes'),
        '_note': 'round 6 kernels: the backward recurrence hands off dh (one plane on the exchange ring instead of three) and loads '
                 'three coefficient planes with plain cached loads; the forward launch of a training pass writes those planes.  '
                 'xcd_replicated_floor: what this design cannot go below with eight non-coherent L2s -- the ring payload AND the '
                 'coefficient planes are fetched into every XCD (8 x 4 planes), payload + canary written once each, 10 planes of '
                 'saved activations.',
    }
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
