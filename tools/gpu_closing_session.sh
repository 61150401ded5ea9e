#!/bin/bash
# Closing session of a round: the whole GPU suite (parity errors recorded), smoke, the driver's bench command, its rocprofv3
# kernel stats, the per-layer table, the step timeline and the PMC traffic passes.  Copies nothing: see tools/collect_profiles.sh.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/closing
rm -rf $O; mkdir -p $O
R=$PWD
export LIDIFF_PARITY_LOG=$O/parity_errors.jsonl
timeout 1700 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
unset LIDIFF_PARITY_LOG
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --no-cpu-baseline --no-train --no-alt --no-coords-roofline --layer-table $O/layer_table.txt > $O/bench_layers.json 2> /dev/null
timeout 300 python tools/debug/step_timeline.py 2>&1 | grep -v amdgpu > $O/step_timeline.txt
timeout 300 python tools/debug/boundary_probe.py 2>&1 | grep -v amdgpu > $O/step_boundary.txt
# row kernels against the tile kernel (flag 8 = LIDIFF_CONV_TILE_ONLY): identity maps, centre + tail, transposed maps
{
  C=""; for s in "0,96,96" "0,128,96" "0,32,32" "1,32,64" "2,64,64" "2,64,128" "2,192,128" "3,128,128" "3,128,256"; do C="$C$s,k1,0,0;$s,k1,0,8;"; done
  echo "# kernel_size 1 (identity map), CFG pair stacked: python tools/conv_probe.py --replicas 2 --cases ... (flags 0 = row kernel, 8 = tile kernel)"
  timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --cases "${C%;}" 2>&1 | grep -v amdgpu
  C=""; for s in "0,96,96" "0,128,96" "0,32,32" "1,32,32" "1,96,96"; do C="$C$s,k3,1,0;$s,k3,1,8;"; done
  echo "# kernel_size 3 as centre + tail (--centre-tail): tail pass + centre pass"
  timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --centre-tail --cases "${C%;}" 2>&1 | grep -v amdgpu
  C=""; for s in "3,256,256" "2,256,128" "2,128,128" "1,128,96" "0,96,96"; do C="$C$s,up,0,0;$s,up,0,8;"; done
  echo "# transposed kernel_size 2 / stride 2 (--up-ordered): pair-list kernel vs tile kernel with offset-grouped rows"
  timeout 600 python tools/conv_probe.py --replicas 2 --iters 30 --up-ordered --cases "${C%;}" 2>&1 | grep -v amdgpu
  echo "# the stem (3 -> 32, one replica): thin-input kernel vs tile kernel, and vs centre + tail on the tile kernel"
  timeout 300 python tools/conv_probe.py --replicas 1 --iters 30 --cases "0,3,32,k3,1,0;0,3,32,k3,1,8" 2>&1 | grep -v amdgpu
  timeout 300 python tools/conv_probe.py --replicas 1 --iters 30 --centre-tail --cases "0,3,32,k3,1,8" 2>&1 | grep -v amdgpu
} > $O/row_kernel_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline --no-train > $O/bench_prof.json 2> $O/bench_prof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
{ echo "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline --no-train (the driver's step count; 12 steps incl. warm-up)"; echo; python tools/rocpd_stats.py $DB --top 60; } > $O/kernel_stats.md 2>&1
python tools/rocpd_gaps.py $DB --min-us 10 --top 15 --last-ms 300 > $O/idle_gaps.txt 2>&1
rm -rf $O/prof
bash tools/pmc_bench.sh > $O/pmc.txt 2>&1
cp gpurun_out/pmc_bench/traffic.json $O/pmc_traffic.json
rm -rf gpurun_out/pmc_bench/FETCH_SIZE gpurun_out/pmc_bench/WRITE_SIZE
tail -3 $O/pytest.txt; cat $O/smoke.txt | tail -1; cut -c1-300 $O/bench_default.json; head -8 $O/kernel_stats.md | cut -c1-150; head -3 $O/step_timeline.txt; head -2 $O/idle_gaps.txt
