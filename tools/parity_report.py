#!/usr/bin/env python
"""gpurun_out/.../parity_errors.jsonl (written by the GPU tests through conftest.record_parity) -> a readable summary:
    python tools/parity_report.py gpurun_out/s1/parity_errors.jsonl > profiles/r03_parity_errors.txt"""
import json
import sys


def main(path):
    rows = [json.loads(l) for l in open(path) if l.strip()]
    by = {}
    for r in rows:
        by.setdefault(r["test"], []).append(r)
    print(f"# achieved parity errors on the MI355X ({len(rows)} records from {path}); bars in the tests are <= 10x these")
    for name, rs in by.items():
        if name == "conv_layer_on_bench_maps":
            w = max(rs, key=lambda r: r["worst_tolerance_fraction"])
            print(f"\n## {name}: {len(rs)} (sigma, level, kind, channels, replica) cases vs the float64 oracle")
            print(f"worst max|err| {max(r['max_abs_err'] for r in rs):.3e}; worst case relative to its bar "
                  f"({w['worst_tolerance_fraction']:.3f} of rtol = atol): sigma {w['sigma']} level {int(w['level'])} {w['kind']} "
                  f"{int(w['c_in'])}->{int(w['c_out'])} hint {int(w['hint'])}: max|err| {w['max_abs_err']:.3e} on outputs up to {w['max_abs_out']:.2f}")
            for r in sorted(rs, key=lambda r: (r["sigma"], r["level"], r["kind"], r["c_in"], r["c_out"], r["replica"])):
                print(f"  sigma {r['sigma']:<4} level {int(r['level'])} {r['kind']:<4} {int(r['c_in']):>3}->{int(r['c_out']):<3} hint {int(r['hint'])} "
                      f"replica {int(r['replica'])}: max|err| {r['max_abs_err']:.2e}  max|out| {r['max_abs_out']:.2f}")
        elif name == "conv_layer_on_bench_maps_split3":
            print(f"\n## {name}: the layers the fused plan runs on the split-operand kernel (rows sorted by neighbour sets), same inputs, same "
                  f"float64 oracle, same bar as the native fp32 kernel: max|err| split3 / native")
            print(f"worst split3 max|err| {max(r['max_abs_err'] for r in rs):.3e} (native on the same cases {max(r['max_abs_err_native'] for r in rs):.3e}); "
                  f"worst ratio split3 / native {max(r['max_abs_err'] / max(r['max_abs_err_native'], 1e-12) for r in rs):.2f}; "
                  f"worst fraction of the bar {max(r['worst_tolerance_fraction'] for r in rs):.3f}")
            for r in sorted(rs, key=lambda r: (r["sigma"], r["level"], r["c_in"], r["c_out"], r["replica"])):
                print(f"  sigma {r['sigma']:<4} level {int(r['level'])} {int(r['c_in']):>3}->{int(r['c_out']):<3} replica {int(r['replica'])}: "
                      f"split3 {r['max_abs_err']:.2e}  native {r['max_abs_err_native']:.2e}  max|out| {r['max_abs_out']:.2f}")
        elif name == "bf16_block":
            print(f"\n## {name}: bf16 training blocks, device vs oracle emulation (float64 sums) from identical inputs")
            for r in rs:
                extra = f" ({int(r['n_params'])} parameter gradients, worst shown)" if "n_params" in r else ""
                print(f"  {r['block']:<7} {r['what']:<34} within 1e-4: {100 * r['within_1e-4']:6.2f} %  worst/scale {r['worst_over_scale']:.2e}  "
                      f"1-cos {1 - r['cosine']:.2e}{extra}")
        else:
            print(f"\n## {name}")
            for r in rs:
                print("  " + ", ".join(f"{k} {v:.3e}" if isinstance(v, float) else f"{k} {v}" for k, v in r.items() if k != "test"))


if __name__ == "__main__":
    main(sys.argv[1])
