#!/bin/bash
# MFMA utilisation of the sparse-conv kernels per layer class (north_star: "MFMA utilisation (feature GEMM) against gfx950 peak"):
# rocprofv3 PMC passes (separate runs, --kernel-trace only -- MI355X_MICROARCH.md "rocprofv3 PMC slots") over tools/conv_probe.py, one
# process per layer so that a kernel instantiation's counters belong to ONE shape.  CFG pair stacked (replicas 2), bench scan at sigma 1.
#   usage: bash tools/pmc_mfma.sh            -> gpurun_out/pmc_mfma/summary.txt (copy to profiles/rNN_pmc_mfma.txt)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_mfma
mkdir -p $OUT
: > $OUT/raw.txt
# tag | conv_probe arguments
CASES=(
 "s8_256_256_k27_split3|--level 3 --cin 256 --cout 256 --kernel split3 --flags 1"
 "s16_256_256_k27_split3|--level 4 --cin 256 --cout 256 --kernel split3 --flags 1"
 "s4_128_128_k27_split3|--level 2 --cin 128 --cout 128 --kernel split3 --flags 1"
 "s8_256_256_k27_split_f16x2|--level 3 --cin 256 --cout 256 --kernel split3 --pieces 2 --flags 1"
 "s4_128_128_k27_split_f16x2|--level 2 --cin 128 --cout 128 --kernel split3 --pieces 2 --flags 1"
 "s8_256_256_k27|--level 3 --cin 256 --cout 256"
 "s8_128_128_k27|--level 3 --cin 128 --cout 128"
 "s4_128_128_k27|--level 2 --cin 128 --cout 128"
 "s4_64_64_k27|--level 2 --cin 64 --cout 64"
 "s1_96_96_k1_rows|--level 0 --cin 96 --cout 96 --kind k1"
 "s8_256_256_k27_bf16_wide|--level 3 --cin 256 --cout 256 --kernel bf16 --rows16"
 "s4_128_128_k27_bf16_wide|--level 2 --cin 128 --cout 128 --kernel bf16 --rows16"
)
for c in "${CASES[@]}"; do
  tag=${c%%|*}; args=${c#*|}
  PASSES="sq1 sq2" bash $R/tools/pmc_probe.sh mfma_$tag $args --replicas 2 --iters 5 2>&1 | grep "^sq" | sed "s/^/$tag /" >> $OUT/raw.txt
  python $R/tools/conv_probe.py $args --replicas 2 --iters 20 2>/dev/null | grep TFLOP | sed "s/^/$tag time /" >> $OUT/raw.txt
done
python - <<PY
import ast, re
rows = {}
for line in open("$OUT/raw.txt"):
    tag, kind, rest = line.split(" ", 2)
    d = rows.setdefault(tag, {})
    if kind == "time":
        d["us"] = float(re.search(r"avg_us=([\d.]+)", rest).group(1)); d["tflops"] = float(re.search(r"TFLOP/s=([\d.]+)", rest).group(1))
        d["pairs"] = int(re.search(r"pairs=(\d+)", rest).group(1)); d["nbrs"] = float(re.search(r"nbrs/row=([\d.]+)", rest).group(1))
        continue
    name = rest.split("(")[0].replace("void lidiff::", "").strip()
    cnt = ast.literal_eval(rest[rest.index("{"):rest.rindex("}") + 1])
    k = d.setdefault("kernels", {}).setdefault(name, {})
    k.update(cnt)
with open("$OUT/summary.txt", "w") as f:
    f.write("# rocprofv3 PMC (tools/pmc_mfma.sh): per launch, summed over the chip; bench scan sigma = 1, CFG pair stacked (replicas 2)\n"
            "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); valu/mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA;\n"
            "# wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of wave time waiting for an instruction's operands / issue); lds_conf = "
            "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; time / TFLOP/s from an unprofiled run of the same case\n")
    for tag, d in rows.items():
        f.write(f"{tag}: {d.get('us', 0):.1f} us  {d.get('tflops', 0):.1f} TFLOP/s  pairs {d.get('pairs', 0)}  nbrs/row {d.get('nbrs', 0):.2f}\n")
        for name, c in d.get("kernels", {}).items():
            if "SQ_INSTS_MFMA" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
                f.write(f"    {name}: incomplete {c}\n"); continue
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
            f.write(f"    {name}: mfma_busy {100 * busy:.1f} %  valu/mfma {(c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / max(1, c['SQ_INSTS_MFMA']):.2f}  "
                    f"wait {100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.0f} %  lds_conf {100 * c['SQ_LDS_BANK_CONFLICT'] / max(1, c['SQ_LDS_IDX_ACTIVE']):.1f} %  "
                    f"insts: mfma {c['SQ_INSTS_MFMA']} valu {c['SQ_INSTS_VALU']} lds {c['SQ_INSTS_LDS']} vmem_rd {c['SQ_INSTS_VMEM_RD']} salu {c['SQ_INSTS_SALU']}  waves {c['SQ_WAVES']}  gui_active {c['GRBM_GUI_ACTIVE']}\n")
import json
js = {"command": "tools/pmc_mfma.sh: rocprofv3 --kernel-trace --pmc (separate passes) over tools/conv_probe.py, one process per layer, "
                 "bench scan sigma 1, replicas 2", "layers": {}}
for tag, d in rows.items():
    for name, c in d.get("kernels", {}).items():
        if "SQ_INSTS_MFMA" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            js["layers"][tag] = {"kernel": name, "us": d.get("us"), "tflops": d.get("tflops"),
                                 "mfma_busy": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024),
                                 "valu_per_mfma": (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / max(1, c["SQ_INSTS_MFMA"]),
                                 "wait_share": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                 "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(1, c["SQ_LDS_IDX_ACTIVE"])}
json.dump(js, open("$OUT/summary.json", "w"), indent=1)
print(open("$OUT/summary.txt").read())
PY
