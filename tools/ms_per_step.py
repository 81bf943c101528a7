"""stdin: the JSON line of bench.py -> its ms_per_step (for A/B loops in gpurun command lines)."""
import json
import sys

line = [l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
r = d.get('roofline') or {}
print(f"{d['ms_per_step']:.3f} ms  {d['value']:.2f} img/s  roofline {r.get('achieved')} {r.get('unit')}")
