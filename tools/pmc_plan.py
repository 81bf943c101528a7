#!/usr/bin/env python3
"""Validate one rocprofv3 --pmc pass against the counter blocks of gfx950 BEFORE it is run.

Why: in round 2 passes that put cache-side counters (TCC_* / TCP_* / TA_* / TD_*) next to a full set of SQ_* counters
never returned on this pool (the profiler had to be killed).  Every pass that stays inside the per-block slot limits of
MI355X_MICROARCH.md ("rocprofv3 PMC slots": SQ 8, TCC 4, GRBM 2) and asks for ONE cache-side block has completed, here
and in round 3 (profiles/r03_pmc_*.txt).  The failing sets over-subscribed a block once their derived counters were
expanded -- e.g. FETCH_SIZE is three TCC counters and WRITE_SIZE two (five for a block of four), a *_sum counter is its
base counter on all 16 x 8 TCC instances -- and rocprofv3 on ROCm 7.2 neither rejects nor multiplexes such a request on
gfx950 (no gfx950 section in its counter definitions), it waits for counters that are never programmed.  So the rule is
enforced on our side: expand every requested counter to hardware counters (tools/pmc_counters_gfx950.json = `rocprofv3
-L` of the box), count per block, refuse a pass that exceeds a block or mixes cache-side blocks with each other or with
more than two SQ counters.

usage: pmc_plan.py COUNTER...      exit 0 and a one-line summary when the pass is fine, exit 2 with the reason otherwise."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SLOTS = {'SQ': 8, 'TCC': 4, 'GRBM': 2, 'TCP': 4, 'TA': 2, 'TD': 2, 'SPI': 2, 'CPC': 2, 'CPF': 2, 'TCA': 2}
CACHE_SIDE = ('TCC', 'TCP', 'TA', 'TD', 'TCA')


def table():
    return json.load(open(os.path.join(HERE, 'pmc_counters_gfx950.json')))['counters']


def expand(name, tab, seen=None):
    """hardware counters {name: block} behind a (possibly derived) counter"""
    seen = seen if seen is not None else set()
    if name in seen:
        return {}
    seen.add(name)
    e = tab.get(name)
    if e is None:
        raise KeyError(name)
    if 'block' in e:
        return {name: e['block']}
    out = {}
    for tok in set(re.findall(r'[A-Za-z_][A-Za-z0-9_]*', e['expr'])):
        if tok in tab:
            out.update(expand(tok, tab, seen))
    return out


def check(counters, tab=None):
    tab = tab or table()
    hw = {}
    for c in counters:
        try:
            hw.update(expand(c, tab))
        except KeyError:
            return False, f'{c}: not a gfx950 counter (rocprofv3 -L)', {}
    per = {}
    for n, b in hw.items():
        per.setdefault(b, []).append(n)
    for b, names in per.items():
        if len(names) > SLOTS.get(b, 2):
            return False, f'block {b}: {len(names)} hardware counters ({", ".join(sorted(names))}) for {SLOTS.get(b, 2)} slots', per
    cache = [b for b in per if b in CACHE_SIDE]
    if len(cache) > 1:
        return False, f'cache-side blocks {cache} in one pass: give each its own pass', per
    if cache and len(per.get('SQ', [])) > 2:
        return False, f'{cache[0]} counters next to {len(per["SQ"])} SQ counters: the combination that hung in round 2', per
    return True, ' '.join(f'{b}:{len(n)}/{SLOTS.get(b, 2)}' for b, n in sorted(per.items())), per


if __name__ == '__main__':
    ok, msg, _ = check(sys.argv[1:])
    print(('ok  ' if ok else 'REFUSED  ') + msg)
    sys.exit(0 if ok else 2)
