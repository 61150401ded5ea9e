#!/bin/bash
# The closing GPU session of a round: the whole GPU suite with the achieved parity errors (default mode and the opt-in two-piece fp16
# mode), the driver's bench command with the per-layer table, its A/B twins (host read per pyramid, one coordinate chain, cached condition
# encoders), the rocprofv3 kernel statistics / queue gaps / step-boundary window of that command, the host-lead probe and the GPU-clock
# marks of a step (no profiler), the PMC passes (traffic on the driver's command, MFMA occupancy per layer class), the split-operand
# kernel's per-layer table / ablations / timeline, the full-trajectory and whole-scan bench forms, the training step by kernel class.
#   usage: bash tools/gpu_final.sh [tag]     -> gpurun_out/<tag>/ (copy what is to be judged into profiles/)
T=${1:-final}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$T; mkdir -p $O
cd $R
LIDIFF_PARITY_LOG=$O/parity_errors.jsonl timeout 2400 python -m pytest tests -m gpu -q -rs > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tools/parity_report.py $O/parity_errors.jsonl > $O/parity_errors.txt 2>&1
LIDIFF_SPLIT_PIECES=2 LIDIFF_PARITY_LOG=$O/parity_errors_f16x2.jsonl timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_f16x2.log 2>&1; tail -2 $O/pytest_gpu_f16x2.log
python tools/parity_report.py $O/parity_errors_f16x2.jsonl > $O/parity_errors_f16x2.txt 2>&1
python bench.py --steps 20 --warmup 5 --layer-table $O/layer_table.txt > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
NOEV="--no-kernel-events --no-cpu-baseline --no-train --no-alt --no-closed-loop --no-coords-roofline"
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 $NOEV > $O/bench_no_events_$i.json 2>> $O/bench_default.err
  LIDIFF_READ_FREE=0 python bench.py --steps 20 --warmup 5 $NOEV > $O/bench_no_events_host_reads_$i.json 2>> $O/bench_default.err
  python bench.py --steps 20 --warmup 5 $NOEV --cached-condition > $O/bench_no_events_cached_$i.json 2>> $O/bench_default.err
done
LIDIFF_PYRAMID_LANES=0 python bench.py --steps 20 --warmup 5 $NOEV > $O/bench_no_events_one_chain.json 2>> $O/bench_default.err
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-train --no-closed-loop > $O/bench_steps50.json 2>> $O/bench_default.err
python bench.py --pipeline --scans 2 > $O/bench_pipeline.json 2>> $O/bench_default.err
for f in $O/bench_no_events*.json $O/bench_steps50.json $O/bench_pipeline.json; do echo "$(basename $f): $(cut -c50-130 $f)"; done
python tools/split3_table.py --f16x2 2>&1 | grep -v amdgpu.ids > $O/split3_table.txt
for L in "3 256 256" "2 128 128"; do set -- $L
  for A in 0 1 2 3 8 16 24 26 27 31 4; do LIDIFF_S3_ABLATE=$A python tools/debug/s3_ablate.py --level $1 --cin $2 --cout $3 2>&1 | grep -v amdgpu.ids | tail -1; done
  python tools/debug/s3_timeline.py --level $1 --cin $2 --cout $3 2>&1 | grep -v amdgpu.ids | tail -3
done > $O/split3_ablation.txt
python tools/micro/gemm_ceiling.py 2>&1 | grep -v amdgpu.ids > $O/gemm_ceiling.txt
python tools/debug/lead_probe.py 2>&1 | grep -v amdgpu.ids > $O/host_lead_probe.txt
python tools/debug/unet_marks.py 2>&1 | grep -v amdgpu.ids > $O/step_marks_gpu_clock.txt
python tools/host_profile.py --steps 10 2>&1 | grep -v amdgpu.ids | head -40 > $O/host_profile.txt
bash tools/gpu_window.sh $T/window > /dev/null 2>&1
for f in rocprofv3_kernel_stats.md idle_gaps.txt main_queue.txt step_boundary_window.txt bench_under_rocprof.json; do cp $O/window/$f $O/ 2>/dev/null; done
bash tools/gpu_train_profile.sh $T/train bf16 > /dev/null 2>&1
bash tools/gpu_train_profile.sh $T/train32 32 > /dev/null 2>&1
cp $O/train/train_kernel_classes_bf16.txt $O/train/train_kernel_stats_bf16.md $O/train/train_kernel_classes_bf16.json $O/ 2>/dev/null
cp $O/train32/train_kernel_classes_32.txt $O/train32/train_kernel_classes_32.json $O/ 2>/dev/null
PMC_STEPS=20 PMC_WARMUP=5 bash tools/pmc_bench.sh > $O/pmc_bench.log 2>&1; cp gpurun_out/pmc_bench/traffic.json $O/pmc_traffic.json
bash tools/pmc_mfma.sh > $O/pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma/summary.txt $O/pmc_mfma.txt; cp gpurun_out/pmc_mfma/summary.json $O/pmc_mfma.json
ls $O; head -8 $O/main_queue.txt
