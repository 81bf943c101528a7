#!/bin/bash
# Sweep of the work-unit targets of the weight-gradient kernels' split-K plans (conv_wgrad.hip: choose / x3h_splits) on the headline
# step: time of the weight-gradient entry points (main kernel + reduce pass) per step and the step time.
# usage: bash tools/sweep_wgrad_units.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
for v in "" "HTD_WGRAD_X3H_UNITS=384" "HTD_WGRAD_X3H_UNITS=512" "HTD_WGRAD_X3H_UNITS=1024" "HTD_WGRAD_UNITS=768" "HTD_WGRAD_UNITS=1152" "HTD_WGRAD_UNITS=1536" "HTD_WGRAD_UNITS=1152 HTD_WGRAD_X3H_UNITS=512" ""; do
  out=$(env HTD_OVERLAP_WGRAD=0 $v timeout -k 10 200 python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --trained-like-steps 0 --profile-kernels 2>&1) || exit 1
  w=$(echo "$out" | grep "^# htd_conv2d_bwd_weight_h2 " | sed 's/.*total= *\([0-9.]*\) ms.*/\1/')
  w2=$(echo "$out" | grep "^# htd_conv2d_bwd_weight " | sed 's/.*total= *\([0-9.]*\) ms.*/\1/')
  r=$(echo "$out" | grep '^{' | python3 $R/tools/ms_per_step.py)
  echo "[$v] bwd_weight_h2 $w ms  bwd_weight $w2 ms (12 steps, weight gradients on the main stream) | $r"
done
