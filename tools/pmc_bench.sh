#!/bin/bash
# HBM traffic of the dominant bench kernel from rocprofv3 PMC passes over bench.py (separate passes,
# --kernel-trace only: MI355X_MICROARCH.md "HBM", "rocprofv3 PMC slots").  Writes gpurun_out/pmc_bench/traffic.json;
# copy it to profiles/r01_pmc_traffic.json to have bench.py report it as roofline.traffic.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$pass -o p -- \
    python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-events --no-alt --no-coords-roofline > $OUT/$pass.log 2>&1
done
cd $R
python - <<PY
import csv, glob, json, os, collections
out = "$OUT"
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(name, "no csv:", open(os.path.join(out, name + ".log")).read()[-500:]); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != name or "spconv_fwd_kernel" not in row["Kernel_Name"]: continue
        a = agg[row["Kernel_Name"].split("(")[0]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
    res[name] = {k: {"sum_kb": v[0], "launches": v[1]} for k, v in agg.items()}
dom = max(res.get("FETCH_SIZE", {}), key=lambda k: res["FETCH_SIZE"][k]["sum_kb"], default=None)
if dom:
    f, w = res["FETCH_SIZE"][dom], res.get("WRITE_SIZE", {}).get(dom, {"sum_kb": 0.0, "launches": 1})
    js = {"kernel": dom, "launches": f["launches"],
          "fetch_bytes_per_launch_raw": 1024.0 * f["sum_kb"] / f["launches"],
          "fetch_bytes_per_launch": 2 * 1024.0 * f["sum_kb"] / f["launches"],     # gfx950: FETCH_SIZE reports 1/2 of wide reads
          "write_bytes_per_launch": 1024.0 * w["sum_kb"] / max(1, w["launches"]),
          "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB), separate passes over bench.py --steps 4 --warmup 1; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); WRITE_SIZE uncalibrated"}
    js["traffic_bytes_per_launch"] = js["fetch_bytes_per_launch"] + js["write_bytes_per_launch"]
    json.dump(js, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps(js, indent=1))
for name, d in res.items():
    for k, v in d.items(): print(name, k[:70], v)
PY
