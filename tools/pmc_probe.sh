#!/bin/bash
# PMC passes over tools/conv_probe.py (one conv shape).  usage: tools/pmc_probe.sh <tag> <conv_probe args...>
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/conv_probe.py "${ARGS[@]}" > $OUT/$name.log 2>&1
}
ARGS=("$@")
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $R
python - <<PY
import csv, glob, collections, os
out="$OUT"
for name in ("sq1","sq2","tcc1","tcc2"):
    files=glob.glob(os.path.join(out,name,"**","*counter_collection.csv"),recursive=True)
    if not files:
        print(name,"no csv; log tail:"); print(open(os.path.join(out,name+".log")).read()[-600:]); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k=row["Kernel_Name"]
        if "spconv" not in k: continue
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); 
    for k,v in agg.items():
        print(name,k[:60],{a:round(b) for a,b in v.items()})
PY
