#!/bin/bash
# Evidence runs of a round (MI355X box, through gpurun): bench lines, rocprofv3 kernel stats and PMC passes.
# usage: bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/*
set -o pipefail
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
echo "== bench lines"
$B --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
HTD_CONV_MATH=0 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_fp32mfma.json 2>/dev/null
HTD_CONV_H2=0 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_bf16x6.json 2>/dev/null          # round 3's arithmetic everywhere
$B --steps 20 --warmup 5 --no-cpu-baseline --trained-like --trained-like-steps 0 > $OUT/bench_trained_like.json 2>/dev/null
$B --steps 20 --warmup 5 --no-cpu-baseline --depth 101 > $OUT/bench_r101.json 2>/dev/null
$B --steps 20 --warmup 5 --no-cpu-baseline --depth 101 --bf16 > $OUT/bench_r101_bf16.json 2>/dev/null
$B --steps 20 --warmup 5 --no-cpu-baseline --depth 101 --dcn > $OUT/bench_r101_dcn.json 2>/dev/null
$B --steps 20 --warmup 5 --no-cpu-baseline --depth 101 --dcn --bf16 > $OUT/bench_r101_dcn_bf16.json 2>/dev/null
$B --steps 5 --warmup 2 --infer --depth 101 --batch 64 > $OUT/bench_infer_r101_b64.json 2>/dev/null
$B --steps 5 --warmup 2 --infer --depth 101 --batch 64 --bf16 > $OUT/bench_infer_r101_b64_bf16.json 2>/dev/null
echo "== kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG/a -o a -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --trained-like-steps 0 > $OUT/prof_train.log 2>&1
head -61 /tmp/p_$TAG/a/*kernel_stats.csv > $OUT/kernel_stats_top60.csv
# the same with the weight gradients on the main stream, as in the steps bench.py brackets with events (overlapped kernels share CUs and each takes longer)
HTD_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG/a0 -o a0 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --trained-like-steps 0 > $OUT/prof_train_no_overlap.log 2>&1
head -61 /tmp/p_$TAG/a0/*kernel_stats.csv > $OUT/kernel_stats_no_overlap_top60.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG/b -o b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --trained-like-steps 0 --depth 101 --bf16 > $OUT/prof_bf16.log 2>&1
head -61 /tmp/p_$TAG/b/*kernel_stats.csv > $OUT/kernel_stats_r101_bf16_top60.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG/c -o c -- python3 $R/bench.py --steps 3 --warmup 1 --infer --depth 101 --batch 64 > $OUT/prof_infer.log 2>&1
head -61 /tmp/p_$TAG/c/*kernel_stats.csv > $OUT/kernel_stats_infer_r101_b64_top60.csv
python3 - /tmp/p_$TAG > $OUT/kernel_stats_totals.txt <<'PY'
import csv, glob, sys
for tag, name, steps in (('a', 'HTD-R50 fp32 train step', 13), ('b', 'HTD-R101 bf16 train step', 13), ('c', 'HTD-R101 fp32 inference B=64', 4)):
    for f in glob.glob(f'{sys.argv[1]}/{tag}/*kernel_stats.csv'):
        rows = list(csv.DictReader(open(f)))
        calls = sum(int(r['Calls']) for r in rows)
        ns = sum(float(r['TotalDurationNs']) for r in rows)
        short = [(int(r['Calls']), float(r['TotalDurationNs'])) for r in rows if float(r['AverageNs']) < 20000]
        print(f'{name}: {steps} steps (warm-up included), {len(rows)} kernels, {calls / steps:.0f} launches and {ns / steps / 1e6:.2f} ms of kernel time per step; '
              f'kernels averaging < 20 us: {sum(c for c, _ in short) / steps:.0f} launches, {sum(t for _, t in short) / steps / 1e6:.2f} ms per step')
PY
echo "== PMC passes (each alone; every set validated by tools/pmc_plan.py)"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do python3 $R/tools/pmc_plan.py $set || exit 2; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_$TAG/f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --trained-like-steps 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_$TAG/w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --trained-like-steps 0 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py /tmp/p_$TAG/f/*counter_collection.csv /tmp/p_$TAG/w/*counter_collection.csv $OUT/hbm_traffic.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/p_$TAG/m -o m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --trained-like-steps 0 > /dev/null 2>&1
python3 $R/tools/pmc_mfma.py /tmp/p_$TAG/m/*counter_collection.csv $OUT/mfma_busy.json > /dev/null
echo "== per-layer convolution tables (stand-alone, warm device)"
python3 $R/tools/bench_conv.py > $OUT/bench_conv_new.log 2>&1
HTD_X3P=0 HTD_WGRAD_X3H=0 HTD_WGRAD_X3D=0 python3 $R/tools/bench_conv.py > $OUT/bench_conv_r02_kernels.log 2>&1
python3 $R/tools/bench_wgrad.py --all > $OUT/wgrad_new.txt 2>&1
HTD_WGRAD_X3D=0 python3 $R/tools/bench_wgrad.py --all > $OUT/wgrad_phased.txt 2>&1
echo "== RoIAlign"
python3 $R/tools/bench_roi_align.py 2048 4 > $OUT/roi_align_2048.log 2>&1
python3 $R/tools/bench_roi_align.py 32768 4 > $OUT/roi_align_32768.log 2>&1
python3 $R/tools/bench_roi_align.py 512 4 --ba > $OUT/roi_align_ba_512.log 2>&1
echo "== round 4: plane-fed 1x1 layers, bf16 LDS-DMA kernel"
HTD_X3P_TUNE=1 python3 $R/tools/bench_planes.py > $OUT/bench_planes.log 2>&1
HTD_BF16Q_TUNE=1 python3 $R/tools/bench_conv_bf16.py > $OUT/bench_conv_bf16q.log 2>&1
echo "== round 4: H2 arithmetic"
$R/tools/micro/mfma_split_products.bin > $OUT/mfma_split_products.txt 2>&1
python3 $R/tools/x3_accuracy.py > $OUT/x3_accuracy.log 2>&1
python3 $R/tools/h2_loss_trajectory.py 80 > $OUT/h2_loss_trajectory.txt 2>&1
python3 $R/tools/h2_trace.py > $OUT/h2_trace.txt 2>&1
python3 $R/tools/conv_layers.py 50 110 > $OUT/conv_layers.txt 2>&1
python3 $R/tools/fixture_margin.py > $OUT/fixture_margin.txt 2>&1
python3 $R/tools/summarize_collection.py $OUT > $OUT/summary.txt 2>&1
ls -la $OUT
