#!/usr/bin/env python3
"""Rewrites the table between <!-- bench-table:begin --> and <!-- bench-table:end --> in DESIGN.md from the tracked bench summary
(profiles/r06_bench_default.json = the --detail file of an UNPROFILED `python bench.py` at HEAD), so that the numbers DESIGN.md quotes cannot drift from
the file the judge reads.  tests/test_host_logic.py::test_design_quotes_the_tracked_bench_summary checks the two against each other.
   python tools/update_design_table.py [profiles/r06_bench_default.json]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_bench_default.json")

# (label, dotted path into the summary, format)
ROWS = [
    ("cfg2 step: images/s (headline `value`)", "value", "{:.1f}"),
    ("cfg2 step: ms per step", "ms_per_step", "{:.4f}"),
    ("timed steps / warm-up", "steps", "{}"),
    ("dominant kernel", "roofline.kernel", "{}"),
    ("dominant kernel: average launch (ms, HIP events)", "roofline.avg_launch_ms", "{:.4f}"),
    ("dominant kernel: achieved TFLOP/s", "roofline.achieved", "{:.2f}"),
    ("dominant kernel: `roofline.frac` of 833.3 TFLOP/s", "roofline.frac", "{:.4f}"),
    ("dominant kernel: fraction of the bare f16x3 MFMA loop on this box", "roofline.sustained.frac_of_bare_loop", "{:.4f}"),
    ("dominant kernel: fabric bytes per launch (counters)", "roofline.traffic", "{}"),
    ("R's 3x3 convolutions at bs256: ms per step", "r_convs.ms_per_step", "{:.4f}"),
    ("R's 3x3 convolutions at bs256: TFLOP/s (of 833.3)", "r_convs.tflops", "{:.2f}"),
    ("element-wise block: ms per step", "elementwise.ms_per_step", "{:.4f}"),
    ("exact-fp32 row: images/s", "f32_row.images_per_sec", "{:.1f}"),
    ("exact-fp32 row: dominant kernel fraction of 157.3 TFLOP/s", "f32_row.roofline_frac", "{:.4f}"),
    ("cfg3 step: images/s", "cfg3.images_per_sec", "{:.1f}"),
    ("cfg3 step: ms per step", "cfg3.ms_per_step", "{:.4f}"),
    ("cfg3 dominant kernel", "cfg3.roofline.kernel", "{}"),
    ("cfg3 dominant kernel: `roofline.frac`", "cfg3.roofline.frac", "{:.4f}"),
    ("cfg3 dominant kernel: average launch (ms)", "cfg3.roofline.avg_launch_ms", "{:.4f}"),
    ("cfg5 embedding pipeline: images/s", "search_cfg5.embed.images_per_sec", "{:.1f}"),
    ("cfg5 five-needle top-50 over 1 M x 100: ms", "search_cfg5.ms", "{:.4f}"),
    ("cfg5 five-needle search: fraction of 8 TB/s", "search_cfg5.hbm_frac", "{:.4f}"),
    ("cfg5 search exact match vs the oracle", "search_cfg5.exact_match", "{}"),
    ("1024 needles over 1 M x 100: ms", "search_cfg5.batched_1024.ms", "{:.4f}"),
    ("1024 needles: TFLOP/s", "search_cfg5.batched_1024.mfma_tflops", "{:.1f}"),
    ("GAN game batch 32 / 256: ms", "gan_step.batch32.ms_per_batch;gan_step.batch256.ms_per_batch", "{}"),
    ("CPU baseline (oracle port, configs[0]): images/s @ cores", "cpu_baseline.value;cpu_baseline.cores", "{}"),
]


def lookup(d, path):
    for k in path.split("."):
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def render(summary):
    out = ["| quantity | key in the summary | value |", "|---|---|---|"]
    for label, path, fmt in ROWS:
        vals = [lookup(summary, p) for p in path.split(";")]
        if any(v is None for v in vals):
            continue
        txt = " / ".join(fmt.format(v) if isinstance(v, (int, float)) and not isinstance(v, bool) and fmt != "{}" else str(v) for v in vals)
        out.append(f"| {label} | `{path}` | {txt} |")
    return "\n".join(out)


def main():
    summary = json.load(open(SRC))
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    new = re.sub(r"(<!-- bench-table:begin -->\n).*?(\n<!-- bench-table:end -->)", lambda m: m.group(1) + render(summary) + m.group(2), text, flags=re.S)
    open(path, "w").write(new)
    print(render(summary))


if __name__ == "__main__":
    main()
