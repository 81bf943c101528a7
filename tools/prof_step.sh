#!/bin/bash
# rocprofv3 kernel trace of one bench.py configuration -> gpurun_out/<tag>_kernel_stats.csv (all kernels) + <tag>_top.txt (per step)
# usage: bash tools/prof_step.sh <tag> [bench.py arguments...]     (13 steps traced: 3 warm-up + 10 timed, no trained-like extra steps)
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$TAG -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --trained-like-steps 0 "$@" > $OUT/${TAG}_run.log 2>&1
cp /tmp/ps_$TAG/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 - $OUT/${TAG}_kernel_stats.csv 13 > $OUT/${TAG}_top.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
calls = sum(int(r['Calls']) for r in rows)
ns = sum(float(r['TotalDurationNs']) for r in rows)
short = [(int(r['Calls']), float(r['TotalDurationNs'])) for r in rows if float(r['AverageNs']) < 20000]
print(f'{len(rows)} kernels, {calls / steps:.0f} launches and {ns / steps / 1e6:.2f} ms of kernel time per step; kernels averaging < 20 us: '
      f'{sum(c for c, _ in short) / steps:.0f} launches, {sum(t for _, t in short) / steps / 1e6:.2f} ms per step')
for r in rows[:90]:
    print(f"{r['Name'][:118]:118s} {int(r['Calls']) / steps:7.1f} {float(r['TotalDurationNs']) / steps / 1e3:9.1f}us {float(r['AverageNs']) / 1e3:8.1f}")
PY
tail -2 $OUT/${TAG}_run.log
head -3 $OUT/${TAG}_top.txt
