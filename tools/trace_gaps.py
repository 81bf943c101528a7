#!/usr/bin/env python3
"""Summarise device idle gaps from a rocprofv3 kernel trace csv: where the GPU waits for the host."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3     # fraction of the trace to skip (warm-up)
ev = ev[int(len(ev) * skip):]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.Counter(); cnt = collections.Counter()
last_end, last_name = ev[0][1], ev[0][2]
for s, e, n in ev[1:]:
    if s > last_end:
        key = (last_name[:60], n[:60])
        gaps[key] += s - last_end; cnt[key] += 1
    if e > last_end:
        last_end, last_name = e, n
print(f'span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms idle {(span-busy)/1e6:.2f} ms kernels {len(ev)}')
small = sum(v for k, v in gaps.items() if v / cnt[k] < 10000)
print(f'gaps with mean < 10us: {small/1e6:.2f} ms')
for k, v in gaps.most_common(40):
    print(f'{v/1e6:8.3f} ms n={cnt[k]:5d} mean {v/cnt[k]/1e3:8.1f} us | {k[0]} -> {k[1]}')
