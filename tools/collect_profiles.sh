#!/bin/bash
# Copies the closing session's outputs (gpurun_out/closing, tools/gpu_closing_session.sh) into profiles/ under this round's names.
R=${1:-r03}
S=gpurun_out/closing
cp $S/bench_default.json profiles/${R}_bench_default.json
cp $S/kernel_stats.md profiles/${R}_rocprofv3_kernel_stats_final.md
cp $S/layer_table.txt profiles/${R}_layer_table_final.txt
cp $S/step_timeline.txt profiles/${R}_step_timeline.txt
cp $S/step_boundary.txt profiles/${R}_step_boundary.txt
cp $S/idle_gaps.txt profiles/${R}_idle_gaps_final.txt
cp $S/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $S/pytest.txt profiles/${R}_pytest_gpu_final.txt
cp $S/row_kernel_probe.txt profiles/${R}_row_kernel_probe.txt
python tools/parity_report.py $S/parity_errors.jsonl > profiles/${R}_parity_errors.txt
ls -la profiles | grep ${R}_ | wc -l
