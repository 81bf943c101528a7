#!/bin/bash
# Every pass is checked by tools/pmc_plan.py first (per-block slot limits of gfx950, one cache-side block per pass, no cache-side
# block next to a set of SQ counters): the round-2 passes that never returned mixed TCC_* / TCP_* / TA_* / TD_* counters with a
# full SQ set, i.e. over-subscribed a block once derived counters were expanded (pmc_plan.py's header has the accounting).
# Cache-side counters therefore get passes of their own, below; a refused set stops the script before anything is launched.
# PMC diagnosis of ONE convolution (tools/one_conv.py arguments): several rocprofv3 --pmc passes, each alone, summed over
# the launches of the kernel class.   usage: bash tools/pmc_diag.sh <tag> <one_conv.py args...>     -> gpurun_out/<tag>.txt
TAG=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd /tmp && export TMPDIR=/tmp
SETS=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_CYCLES"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
)
for s in "${SETS[@]}"; do python3 $R/tools/pmc_plan.py $s > /dev/null || { python3 $R/tools/pmc_plan.py $s; exit 2; }; done
rm -rf /tmp/pd_$TAG; i=0; : > $R/gpurun_out/$TAG.progress
for s in "${SETS[@]}"; do
  echo "pass $i: $s" >> $R/gpurun_out/$TAG.progress
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $s --output-format csv -d /tmp/pd_$TAG/$i -o p -- python3 $R/tools/one_conv.py "$@" > /tmp/pd_$TAG.$i.log 2>&1
  rc=$?; echo "  rc=$rc" >> $R/gpurun_out/$TAG.progress
  if [ $rc -ne 0 ]; then tail -5 /tmp/pd_$TAG.$i.log >> $R/gpurun_out/$TAG.progress; exit 1; fi      # no further GPU step after a killed one
  i=$((i+1))
done
python3 - /tmp/pd_$TAG > $R/gpurun_out/$TAG.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        cls = 'conv_igemm' if 'conv_igemm' in k else 'conv_x3p' if 'conv_x3p_kernel' in k else 'conv_wgrad' if 'conv_wgrad' in k else None
        if cls is None: continue
        acc[cls][r['Counter_Name']] += float(r['Counter_Value'])
        n[(cls, r['Counter_Name'])].add(r.get('Dispatch_Id'))
for cls, c in acc.items():
    print(cls)
    for k in sorted(c):
        print('  %-32s %16.0f  per launch %14.0f' % (k, c[k], c[k] / max(1, len(n[(cls, k)]))))
PY
cat $R/gpurun_out/$TAG.txt
