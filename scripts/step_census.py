#!/usr/bin/env python3
"""Dispatch census of ONE steady-state step from a rocprofv3 --kernel-trace CSV.

  python scripts/step_census.py gpurun_out/prof_r04/dcgan/p_kernel_trace.csv [--marker adam_kernel --per-step 2] [--seq]

A step = the dispatches after the last `--marker` launch of the previous step up to and including the last marker launch of this
one (DCGAN: two Adam launches per step, the second one ends the step).  The LAST complete step of the trace is taken (a replayed
graph step unless the bench ran eagerly); --steps N averages the counts over the last N steps.  Unlike the --stats table divided by
the step count this leaves out set-up dispatches (uploads, flattening, calibration)."""
import argparse
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("pcg::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"pcg::TileCfg<(\d+), (\d+), \d+, \d+(?:, (true|false), \d+, \d+(?:, (?:true|false))?)?>",
                  lambda m: f"{m.group(1)}x{m.group(2)}" + ("/swz3" if m.group(3) == "true" else ""), name)
    return name.split("(")[0]


def census(path, marker="adam_kernel", per_step=2, steps=1):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    ends = marks[per_step - 1::per_step]                 # index of the step-ending marker of every step
    if len(ends) < steps + 1:
        raise SystemExit(f"trace has {len(ends)} steps; need {steps + 1}")
    a, b = ends[-steps - 1] + 1, ends[-1] + 1
    seg = rows[a:b]
    names = [short(r["Kernel_Name"]) for r in seg]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg]
    per = collections.OrderedDict()
    for n, d in zip(names, dur):
        e = per.setdefault(n, [0, 0.0])
        e[0] += 1; e[1] += d
    span = (int(seg[-1]["End_Timestamp"]) - int(rows[a - 1]["End_Timestamp"])) / 1e3 / steps
    return {"dispatches_per_step": len(seg) / steps, "busy_us_per_step": sum(dur) / steps, "span_us_per_step": span,
            "by_kernel": {k: {"per_step": v[0] / steps, "avg_us": v[1] / v[0]} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])},
            "sequence": list(zip(names, [round(d, 1) for d in dur])) if steps == 1 else None}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="adam_kernel")
    ap.add_argument("--per-step", type=int, default=2)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--seq", action="store_true", help="print the dispatch sequence of the step")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    c = census(a.trace, a.marker, a.per_step, a.steps)
    print(f"{c['dispatches_per_step']:.1f} dispatches/step, busy {c['busy_us_per_step']:.0f} us, span {c['span_us_per_step']:.0f} us")
    for k, v in c["by_kernel"].items():
        print(f"  {v['per_step']:6.1f} x {v['avg_us']:8.1f} us  {k}")
    if a.seq and c["sequence"]:
        for i, (n, d) in enumerate(c["sequence"]):
            print(f"{i:4d} {d:8.1f}  {n}")
    if a.json:
        json.dump(c, open(a.json, "w"), indent=1)
