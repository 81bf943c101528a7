#!/bin/bash
# Sweep of the constants of conv_x3.hip's work-plan model (plan_x3p) on the headline step: per setting the total time of the
# two x3p entry points over the profiled steps (bench.py --profile-kernels) and the step rate.  usage: bash tools/sweep_x3p_plan.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
for v in "" "HTD_X3P_PLAN_OV=4" "HTD_X3P_PLAN_OV=16" "HTD_X3P_PLAN_LAT1=125 HTD_X3P_PLAN_LAT2=105" "HTD_X3P_PLAN_LAT1=170 HTD_X3P_PLAN_LAT2=125" \
         "HTD_X3P_PLAN_MAXREM=32" "HTD_X3P_PLAN_MAXREM=8" "HTD_X3P_PLAN_LAT1=200 HTD_X3P_PLAN_LAT2=150" ""; do
  out=$(env HTD_X3P_TUNE=1 $v timeout -k 10 200 python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --profile-kernels 2>&1) || exit 1
  f=$(echo "$out" | grep "^# htd_conv2d_fwd_x3p " | sed 's/.*total= *\([0-9.]*\) ms.*/\1/')
  d=$(echo "$out" | grep "^# htd_conv2d_bwd_data_x3p " | sed 's/.*total= *\([0-9.]*\) ms.*/\1/')
  r=$(echo "$out" | grep '^{' | sed 's/.*"value": \([0-9.]*\).*/\1/')
  echo "[$v] fwd_x3p $f ms  bwd_data_x3p $d ms  step $r img/s"
done
