#!/bin/bash
# PMC passes (each validated by tools/pmc_plan.py, each alone) over an arbitrary command, summed over the launches of the
# kernels whose name contains <substring>.   usage: bash tools/pmc_kernel.sh <tag> <substring> <python-script> [args...]
# -> gpurun_out/<tag>.txt   (the program after `--` must be python3 <script> itself: no wrappers under rocprofv3)
TAG=$1; SUB=$2; shift 2
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_SMEM"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"
)
for s in "${SETS[@]}"; do python3 $R/tools/pmc_plan.py $s > /dev/null || { python3 $R/tools/pmc_plan.py $s; exit 2; }; done
rm -rf /tmp/pk_$TAG; i=0; : > $R/gpurun_out/$TAG.progress
for s in "${SETS[@]}"; do
  echo "pass $i: $s" >> $R/gpurun_out/$TAG.progress
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $s --output-format csv -d /tmp/pk_$TAG/$i -o p -- python3 "$@" > /tmp/pk_$TAG.$i.log 2>&1
  rc=$?; echo "  rc=$rc" >> $R/gpurun_out/$TAG.progress
  if [ $rc -ne 0 ]; then tail -5 /tmp/pk_$TAG.$i.log >> $R/gpurun_out/$TAG.progress; exit 1; fi
  i=$((i+1))
done
python3 - /tmp/pk_$TAG "$SUB" > $R/gpurun_out/$TAG.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r['Kernel_Name']:
            continue
        acc[r['Counter_Name']] += float(r['Counter_Value'])
        n[r['Counter_Name']].add(r.get('Dispatch_Id'))
print('kernels containing', repr(sys.argv[2]))
for k in sorted(acc):
    print('  %-34s %18.0f  per launch %16.0f  (%d launches)' % (k, acc[k], acc[k] / max(1, len(n[k])), len(n[k])))
PY
cat $R/gpurun_out/$TAG.txt
