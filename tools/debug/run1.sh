cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma.log 2>&1; tail -25 gpurun_out/pmc_mfma/summary.txt
PMC_STEPS=20 PMC_WARMUP=5 bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; head -8 gpurun_out/pmc_bench.log
