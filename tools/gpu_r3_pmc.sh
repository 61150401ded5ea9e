#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/pmc_bench.sh 2>&1 | tail -30
ls -la gpurun_out/pmc_bench/ | head; rm -rf gpurun_out/pmc_bench/FETCH_SIZE gpurun_out/pmc_bench/WRITE_SIZE
