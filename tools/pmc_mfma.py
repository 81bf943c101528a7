#!/usr/bin/env python3
"""Aggregate a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE
SQ_LDS_BANK_CONFLICT ...) into per-kernel-class matrix-pipe and LDS utilisation.
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
  (MI355X_MICROARCH.md: MFMA_BUSY counts cycles, 32 per v_mfma_f32_32x32x16_bf16; GUI_ACTIVE is summed over the XCDs).
usage: pmc_mfma.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    cls = 'conv_igemm' if 'conv_igemm_kernel' in n else 'conv_x3p' if ('conv_x3p_kernel' in n or 'conv_x3q_kernel' in n) else 'conv_wgrad' if 'conv_wgrad' in n else None
    if cls is None:
        continue
    acc[cls][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r.get('Dispatch_Id') or r.get('Correlation_Id'), n)
    if key not in seen:
        seen.add(key)
        launches[cls] += 1
out = {'note': __doc__.split('usage')[0].strip(), 'kernels': {}}
for cls, c in acc.items():
    cycles = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    d = dict(launches=launches[cls], counters={k: v for k, v in c.items()})
    if cycles > 0:
        d['kernel_cycles_sum'] = cycles
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            d['mfma_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cycles), 4)
        if 'SQ_LDS_IDX_ACTIVE' in c and 'SQ_BUSY_CYCLES' in c:
            d['lds_active_per_sq_busy'] = round(c['SQ_LDS_IDX_ACTIVE'] / c['SQ_BUSY_CYCLES'], 4)
        if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
            d['lds_bank_conflict_frac'] = round(c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'], 4)
    out['kernels'][cls] = d
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != 'counters'} for k, v in out['kernels'].items()}, indent=1))
