#!/bin/bash
# Round 6: matrix-pipe occupancy, LDS conflicts, HBM-side traffic and L2 hit rate of the split-operand kernel (spconv_split3.hip)
# next to the native fp32 kernel on the same maps -- rocprofv3 PMC passes (separate runs, --kernel-trace only) over
# tools/conv_probe.py, one process per layer; bench scan, CFG pair stacked.   usage: bash tools/pmc_split3.sh -> gpurun_out/pmc_split3.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_split3.txt
: > $OUT
CASES=(
 "s8_256_256_split3_sorted|--level 3 --cin 256 --cout 256 --kernel split3 --flags 1"
 "s8_256_256_split3_table_order|--level 3 --cin 256 --cout 256 --kernel split3"
 "s16_256_256_split3_sorted|--level 4 --cin 256 --cout 256 --kernel split3 --flags 1"
 "s4_128_128_split3_sorted|--level 2 --cin 128 --cout 128 --kernel split3 --flags 1"
 "s4_64_64_split3_sorted|--level 2 --cin 64 --cout 64 --kernel split3 --flags 1"
 "s8_256_256_native|--level 3 --cin 256 --cout 256"
 "s4_128_128_native|--level 2 --cin 128 --cout 128"
)
for c in "${CASES[@]}"; do
  tag=${c%%|*}; args=${c#*|}
  echo "== $tag" >> $OUT
  python $R/tools/conv_probe.py $args --replicas 2 --iters 20 2>/dev/null | grep TFLOP >> $OUT
  PASSES="sq1 sq2 tcc1 tcc2" bash $R/tools/pmc_probe.sh s3_$tag $args --replicas 2 --iters 5 2>&1 | grep -E "^(sq|tcc)" >> $OUT
done
python - <<PY
import ast, re
tag = None; rows = {}
for line in open("$OUT"):
    if line.startswith("== "): tag = line[3:].strip(); rows[tag] = {"c": {}}; continue
    if "TFLOP/s=" in line:
        rows[tag]["us"] = float(re.search(r"avg_us=([\d.]+)", line).group(1)); rows[tag]["tf"] = float(re.search(r"TFLOP/s=([\d.]+)", line).group(1)); continue
    if line[:2] in ("sq", "tc") and "{" in line:
        rows[tag]["c"].update(ast.literal_eval(line[line.index("{"):line.rindex("}") + 1]))
with open("$OUT", "a") as f:
    f.write("\n# summary (per launch; FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950 correction; sizes in KB as rocprofv3 reports them)\n")
    for tag, d in rows.items():
        c = d["c"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c: f.write(f"{tag}: incomplete {c}\n"); continue
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
        clk = c["GRBM_GUI_ACTIVE"] / 8 / (d.get("us", 1) * 1e-6) / 1e9
        gb = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / 1e9
        hit = c.get("TCC_HIT_sum", 0) / max(1, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))
        f.write(f"{tag}: {d.get('us', 0):.0f} us  {d.get('tf', 0):.1f} TFLOP/s fp32-equivalent | mfma_busy {100 * busy:.1f} %  clock(under the profiler) {clk:.2f} GHz  "
                f"mfma insts {c['SQ_INSTS_MFMA']:.0f}  valu/mfma {(c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / max(1, c['SQ_INSTS_MFMA']):.2f}  "
                f"wait {100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.0f} %  lds_conf {100 * c['SQ_LDS_BANK_CONFLICT'] / max(1, c['SQ_LDS_IDX_ACTIVE']):.1f} % | "
                f"HBM-side {gb:.2f} GB / launch  L2 hit {100 * hit:.0f} %\n")
print(open("$OUT").read())
PY
