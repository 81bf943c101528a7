#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
for v in "" "HTD_X3P_PLAN_LAUNCH_US=8" "HTD_X3P_PLAN_LAUNCH_US=12" "HTD_X3P_PLAN_LAUNCH_US=20" "HTD_X3P_PLAN_OV=12" "HTD_X3P_PLAN_OV=16 HTD_X3P_PLAN_LAUNCH_US=12" "HTD_X3P_PLAN_MAXREM=8 HTD_X3P_PLAN_LAUNCH_US=12" ""; do
  out=$(env HTD_X3P_TUNE=1 $v timeout -k 10 200 python3 $R/bench.py --steps 16 --warmup 6 --no-cpu-baseline --trained-like-steps 0 --profile-kernels 2>&1) || exit 1
  f=$(echo "$out" | grep "^# htd_conv2d_fwd_x3h " | sed 's/.*total= *\([0-9.]*\) ms.*/\1/')
  d=$(echo "$out" | grep "^# htd_conv2d_bwd_data_x3h " | sed 's/.*total= *\([0-9.]*\) ms.*/\1/')
  r=$(echo "$out" | grep '^{' | python3 $R/tools/ms_per_step.py)
  echo "[$v] fwd_x3h $f ms  bwd_data_x3h $d ms  | $r"
done
