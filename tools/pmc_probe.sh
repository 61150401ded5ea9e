#!/bin/bash
# PMC passes over tools/conv_probe.py (one conv shape).  usage: tools/pmc_probe.sh <tag> <conv_probe args...>
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=("$@")
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/tools/conv_probe.py "${ARGS[@]}" > $OUT/$name.log 2>&1
}
PASSES=${PASSES:-"sq1 sq2 sq3 tcc1 tcc2"}
for pass in $PASSES; do
  case $pass in
    sq1) run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ;;
    sq2) run sq2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE ;;
    sq3) run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH ;;
    tcc1) run tcc1 FETCH_SIZE ;;
    tcc2) run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum ;;
  esac
done
cd $R
python - <<PY
import csv, glob, collections, os
out="$OUT"
for name in "$PASSES".split():
    files=glob.glob(os.path.join(out,name,"**","*counter_collection.csv"),recursive=True)
    if not files:
        print(name,"no csv; log tail:"); print(open(os.path.join(out,name+".log")).read()[-600:]); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for row in csv.DictReader(open(files[0])):
        k=row["Kernel_Name"]
        if "spconv" not in k: continue
        agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[(k,row["Counter_Name"])]+=1
    for k,v in agg.items():
        print(name,k[:70],{a:round(b/n[(k,a)]) for a,b in v.items()}, "(per launch, %d launches)"%max(n[(k,a)] for a in v))
PY
