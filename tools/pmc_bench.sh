#!/bin/bash
# HBM-side traffic of the bench's kernels from rocprofv3 PMC passes over the SAME command the driver runs
# (python bench.py --steps 10 --warmup 2; separate passes per counter, --kernel-trace only: MI355X_MICROARCH.md "HBM",
# "rocprofv3 PMC slots").  Writes gpurun_out/pmc_bench/traffic.json: one entry per lidiff kernel (every spconv_fwd_kernel
# instantiation, the coordinate kernels) with FETCH_SIZE doubled per the guide's gfx950 correction + WRITE_SIZE, per launch.
# Copy it to profiles/rNN_pmc_traffic.json: bench.py reports it as roofline.traffic / roofline_narrow.traffic /
# roofline_hbm.traffic (labelled static: counters cannot be read inside the timed process).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_bench
mkdir -p $OUT
STEPS=${PMC_STEPS:-10}
WARM=${PMC_WARMUP:-2}
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$pass -o p -- \
    python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline --no-train --no-closed-loop > $OUT/$pass.log 2>&1
done
cd $R
python - <<PY
import csv, glob, json, os, collections, re
out = "$OUT"
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(name, "no csv:", open(os.path.join(out, name + ".log")).read()[-500:]); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != name or "lidiff::" not in row["Kernel_Name"]: continue
        a = agg[re.sub(r"^void ", "", row["Kernel_Name"].split("(")[0])]
        a[0] += float(row["Counter_Value"]); a[1] += 1
    res[name] = agg
kernels = {}
for k, (kb, n) in res.get("FETCH_SIZE", {}).items():
    w = res.get("WRITE_SIZE", {}).get(k, [0.0, 1])
    kernels[k] = {"launches": n, "fetch_bytes_per_launch_raw": 1024.0 * kb / n, "fetch_bytes_per_launch": 2048.0 * kb / n,
                  "write_bytes_per_launch": 1024.0 * w[0] / max(1, w[1]),
                  "traffic_bytes_per_launch": 2048.0 * kb / n + 1024.0 * w[0] / max(1, w[1]),
                  "traffic_bytes_total": 2048.0 * kb + 1024.0 * w[0]}
conv = {k: v for k, v in kernels.items() if "spconv_fwd_kernel" in k or "spconv_fwd_split3_kernel" in k}
dom = max(conv, key=lambda k: conv[k]["traffic_bytes_total"], default=None)
js = {"command": f"python bench.py --steps $STEPS --warmup $WARM (all {int('$STEPS') + int('$WARM')} steps counted)",
      "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md "
              "(gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated; L2 -> fabric side: Infinity-Cache hits are included",
      "kernels": kernels}
if dom:
    js.update({"kernel": dom, **{k: conv[dom][k] for k in ("launches", "fetch_bytes_per_launch_raw", "fetch_bytes_per_launch",
                                                           "write_bytes_per_launch", "traffic_bytes_per_launch")}})
json.dump(js, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["traffic_bytes_total"])[:25]:
    print(f"{k[:80]:<80} launches {v['launches']:5d}  traffic/launch {v['traffic_bytes_per_launch'] / 1e6:10.2f} MB")
PY
