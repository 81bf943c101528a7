"""Device idle time of a step from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`).

    python tools/timeline_gaps.py <*_kernel_trace.csv> [--steps N] [--skip S]

The trace's dispatches are merged into busy intervals (kernels of the two streams overlap); what is left between them is
idle time of the device: in-queue dispatch gaps (a few us each) and the places where the queue ran dry because the host was
behind or waited on a read-back.  Prints busy / idle per step, the gap histogram and the longest gaps with the kernels
on both sides, for the last `steps` steps' worth of dispatches (the first `skip` fraction of the trace is warm-up).
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--steps', type=int, default=10, help='timed steps in the traced run (after warm-up)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--top', type=int, default=25)
    args = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(args.trace)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # the timed steps are the last steps / (steps + warmup) of the launches (every step launches the same kernels)
    first = len(rows) * args.warmup // (args.steps + args.warmup)
    rows = rows[first:]
    span = rows[-1][1] - rows[0][0]
    busy = 0
    gaps = []
    cur_s, cur_e, cur_name = rows[0]
    for s, e, name in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_name, name))
            cur_s, cur_e, cur_name = s, e, name
        elif e > cur_e:
            cur_e, cur_name = e, name
    busy += cur_e - cur_s
    n = args.steps
    idle = sum(g for g, _, _ in gaps)
    print(f'{len(rows) / n:.0f} launches per step; span {span / n / 1e6:.2f} ms, busy {busy / n / 1e6:.2f} ms, idle '
          f'{idle / n / 1e6:.2f} ms per step in {len(gaps) / n:.0f} gaps')
    edges = (2e3, 5e3, 10e3, 20e3, 50e3, 100e3, 1e9)
    lo = 0
    for hi in edges:
        sel = [g for g, _, _ in gaps if lo <= g < hi]
        print(f'  gaps {lo / 1e3:6.0f} - {hi / 1e3:8.0f} us: {len(sel) / n:7.1f} per step, {sum(sel) / n / 1e6:6.3f} ms per step')
        lo = hi
    # the long gaps, grouped by the pair of kernels around them
    pairs = {}
    for g, a, b in gaps:
        if g >= 20e3:
            k = (a[:70], b[:70])
            c = pairs.setdefault(k, [0, 0])
            c[0] += 1
            c[1] += g
    print(f'gaps >= 20 us by neighbours (count per step, ms per step):')
    for (a, b), (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print(f'  {c / n:5.1f} {t / n / 1e6:6.3f}  {a}  ->  {b}')


if __name__ == '__main__':
    main()
