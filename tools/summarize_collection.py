"""Summary of a tools/collect_profiles.sh output directory (bench lines, kernel totals, PMC classes, trace agreement).
usage: python tools/summarize_collection.py gpurun_out/r03"""
import csv
import json
import os
import sys

d = sys.argv[1]
for f in ('bench.json', 'bench_bf16x6.json', 'bench_fp32mfma.json', 'bench_trained_like.json', 'bench_r101.json', 'bench_r101_bf16.json', 'bench_r101_dcn.json',
          'bench_r101_dcn_bf16.json', 'bench_infer_r101_b64.json', 'bench_infer_r101_b64_bf16.json'):
    if not os.path.exists(os.path.join(d, f)):
        continue
    b = json.loads([l for l in open(os.path.join(d, f)) if l.startswith('{')][-1])
    r = b.get('roofline') or {}
    print(f"{f:34s} {b['value']:8.2f} img/s {b['ms_per_step']:8.2f} ms  {r.get('kernel', '')[:22]:22s} {r.get('achieved')} {r.get('frac')} "
          f"avg_ms={r.get('avg_launch_ms')} launches={r.get('launches')} traffic={r.get('traffic')}")
print(open(os.path.join(d, 'kernel_stats_totals.txt')).read().strip())
m = json.load(open(os.path.join(d, 'mfma_busy.json')))
for k, v in m['kernels'].items():
    print('mfma_busy', k, v.get('mfma_busy'), 'launches', v.get('launches'))
h = json.load(open(os.path.join(d, 'hbm_traffic.json')))
for k, v in h['kernels'].items():
    print('hbm', k, v.get('hbm_bytes_per_launch_corrected'), v.get('launches'))
for name in ('kernel_stats_no_overlap_top60.csv', 'kernel_stats_top60.csv'):
    rows = list(csv.DictReader(open(os.path.join(d, name))))
    x = [r for r in rows if 'conv_x3p_kernel' in r['Name'] or 'conv_x3q_kernel' in r['Name']]
    calls = sum(int(r['Calls']) for r in x)
    ns = sum(float(r['TotalDurationNs']) for r in x)
    e = [r for r in rows if 'conv_x3p_splitk' in r['Name']]
    ec = sum(int(r['Calls']) for r in e)
    w = [r for r in rows if 'conv_wgrad' in r['Name']]
    g = [r for r in rows if 'conv_igemm' in r['Name']]
    print(f"{name}: x3p {calls} launches, {ns / calls / 1e3:.2f} us avg, {ns / 13e6:.2f} ms/step; reduce passes {ec / 13:.0f}/step at "
          f"{sum(float(r['TotalDurationNs']) for r in e) / max(1, ec) / 1e3:.2f} us; wgrad {sum(float(r['TotalDurationNs']) for r in w) / 13e6:.2f} ms/step; "
          f"igemm {sum(float(r['TotalDurationNs']) for r in g) / 13e6:.2f} ms/step")
