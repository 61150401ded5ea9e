#!/usr/bin/env python
"""Where a TRAINING step's device time goes, by kernel class, from a rocprofv3 --kernel-trace rocpd database over
tools/train_probe.py ("other 50 ms" as named parts):
    python tools/train_classes.py x_results.db --steps 4 [--skip-ms 0] [--json out.json]
Only the last `steps` steps are counted (the trace's tail: --tail-ms, default = everything after the first optimizer launch of the
warm-up is NOT separable by name, so the probe is run with --warmup 1 and the tail share steps / (steps + warmup) of every class's
launches is taken -- launches per step are identical from step to step)."""
import collections
import json
import re
import sqlite3
import sys

CLASSES = [
    ("conv fwd + dX, bf16 operands (spconv_bf16.hip)", r"spconv_fwd_bf16|spconv_bf16"),
    ("conv dW (spconv_bwd_w*, slice reduction)", r"spconv_bwd_w|dw_reduce"),
    ("conv fwd + dX, fp32 tile / row / thin kernels", r"spconv_fwd_kernel|spconv_rows_kernel|spconv_thin"),
    ("weight packing", r"pack_weights"),
    ("BatchNorm (norm.hip)", r"bn_"),
    ("GEMMs (hipBLASLt / rocBLAS: conditioning + head MLPs)", r"Cijk_|gemm|Gemm"),
    ("row gathers / segment sums (slice, conditioning, their backward)", r"gather_rows|segment_sum|gather_mul|gather_bias"),
    ("coordinate maps (hash, kernel maps, rulebooks, voxel mean, matches)", r"insert_kernel|flag_count|scan_write|inverse_kernel|kernel_map|tail_|rb_|mean_|nn_match|coord_max|morton|scan_i32|floor_kernel|points_to_field"),
    ("optimizer (fused Adam)", r"multi_tensor|adam|Adam"),
    ("sorts / scans (torch: scatter CSR)", r"sort|Sort|scan|Scan|searchsorted|radix"),
    ("torch elementwise / reductions / copies", r"at::native|elementwise|reduce_kernel|CatArray|index"),
    ("memset / memcpy (runtime)", r"__amd_rocclr_|fillBuffer|copyBuffer"),
]


def main():
    path = sys.argv[1]
    arg = lambda k, d: type(d)(sys.argv[sys.argv.index(k) + 1]) if k in sys.argv else d
    steps, warmup = arg("--steps", 4), arg("--warmup", 1)
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {name_col}, count(*), sum(end-start) from kernels group by {name_col}").fetchall()
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    agg = collections.OrderedDict((c, [0, 0.0, collections.Counter()]) for c, _ in CLASSES + [("other", "")])
    for name, calls, ns in rows:
        cls = next((c for c, pat in CLASSES if re.search(pat, name)), "other")
        agg[cls][0] += calls
        agg[cls][1] += ns / 1e6
        agg[cls][2][re.sub(r"\(.*$", "", name).replace("void ", "")[:70]] += ns / 1e6
    share = 1.0 / (steps + warmup)            # every step launches the same kernels: one step = 1 / (all steps) of the totals
    total = sum(v[1] for v in agg.values()) * share
    out = {"steps_in_trace": steps + warmup, "device_busy_ms_per_step": total, "classes": {}}
    print(f"# training step by kernel class: {path} ({steps + warmup} steps in the trace; per-step = totals / {steps + warmup}); "
          f"trace span {(span[1] - span[0]) / 1e6:.1f} ms")
    print("class | launches/step | ms/step | % of device-busy | top kernels (ms/step)")
    for cls, (calls, ms, names) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if calls == 0:
            continue
        top = ", ".join(f"{n} {t * share:.2f}" for n, t in names.most_common(3))
        print(f"{cls} | {calls * share:.0f} | {ms * share:.2f} | {100 * ms * share / total:.1f} | {top}")
        out["classes"][cls] = {"launches_per_step": calls * share, "ms_per_step": ms * share}
    print(f"TOTAL device-busy {total:.2f} ms per step (kernel durations summed over all queues; the step's wall time is bench.py's)")
    if "--json" in sys.argv:
        json.dump(out, open(arg("--json", ""), "w"), indent=1)


if __name__ == "__main__":
    main()
