#!/usr/bin/env python3
"""Aggregate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- collected in SEPARATE runs of the same command) into
per-launch HBM bytes per kernel class, with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B
read requests as 64 B: doubled; both counters are in KiB).  usage: pmc_traffic.py <fetch.csv> <write.csv> <out.json>"""
import csv, json, sys, collections


def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        n = r['Kernel_Name']
        cls = 'conv_x3p' if ('conv_x3p_kernel' in n or 'conv_x3q_kernel' in n) else 'conv_igemm' if 'conv_igemm_kernel' in n else 'conv_wgrad' if 'conv_wgrad' in n else \
            'roi_align_fwd' if 'roi_align_kernel<false>' in n else 'roi_align_bwd' if 'roi_align' in n and 'bbox' not in n else None
        if cls is None:
            continue
        acc[cls][0] += 1
        acc[cls][1] += float(r['Counter_Value'])
    return acc


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over the same command (bench.py --steps 2 --warmup 1 unless the file name says otherwise); '
               'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); per-launch '
               'averages over all launches of the kernel class', 'kernels': {}}
for cls in fetch:
    nf, f = fetch[cls]
    nw, w = write.get(cls, [nf, 0.0])
    out['kernels'][cls] = dict(launches=nf, fetch_size_kib_per_launch=round(f / nf, 1),
                               write_size_kib_per_launch=round(w / max(nw, 1), 1),
                               hbm_bytes_per_launch_corrected=int((2 * f / nf + w / max(nw, 1)) * 1024))
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1))
