#!/usr/bin/env python3
"""Turn the output of scripts/collect_profiles.sh (gpurun_out/prof_<tag>/) into the tracked summaries under profiles/:

  python scripts/summarize_round.py r02
writes  <tag>_kernel_stats.md (DCGAN bench), <tag>_{countergan,wgan,house}_kernel_stats.md, <tag>_pmc_traffic.json,
        <tag>_house_launches.json, <tag>_rocprofv3_kernel_stats.csv (raw, DCGAN) and copies the JSON lines (<tag>_*_line.json).
FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced reads: doubled here, as MI355X_MICROARCH.md prescribes;
WRITE_SIZE is exact.  Both are reported by rocprofv3 in KiB."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    import re
    name = name.replace("pcg::(anonymous namespace)::", "").replace("void ", "")
    # TileCfg<BM, BN, WM, WN[, swizzled, min waves/SIMD, prefetch depth]> -> BMxBN (+ "/swz3" for the unpadded three-per-CU config)
    name = re.sub(r"pcg::TileCfg<(\d+), (\d+), \d+, \d+(?:, (true|false), \d+, \d+(?:, (?:true|false))?)?>",
                  lambda m: f"{m.group(1)}x{m.group(2)}" + ("/swz3" if m.group(3) == "true" else ""), name)
    return name.split("(")[0]


def find(sub, pattern):
    hits = glob.glob(os.path.join(src, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def stats_md(sub, dest, title, steps):
    path = find(sub, "*kernel_stats.csv")
    if not path:
        print("missing", sub)
        return
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    with open(os.path.join(out, dest), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {title}, round {tag}\n\n")
        f.write(f"{steps} executed steps in the trace (capture warm-up + warm-up + timed; kernels inside a replayed HIP graph are traced like any other).\n\n")
        f.write("| kernel | launches/step | avg µs | ms/step | % of GPU time |\n|---|---|---|---|---|\n")
        for r in rows:
            t = float(r["TotalDurationNs"])
            if t / tot < 0.0005:
                continue
            f.write(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {t / 1e6 / steps:.3f} | {100 * t / tot:.1f} |\n")
        f.write(f"\nGPU busy time: {tot / 1e6 / steps:.3f} ms/step, {calls / steps:.1f} launches/step\n")
    return path, tot, calls


r = stats_md("dcgan", f"{tag}_kernel_stats.md", "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-calib --no-secondary (DCGAN 64x64, batch 512)", 16)
if r:
    shutil.copy(r[0], os.path.join(out, f"{tag}_rocprofv3_kernel_stats.csv"))
stats_md("countergan", f"{tag}_countergan_kernel_stats.md", "python3 scripts/bench_countergan.py --steps 5 --warmup 2 (batch 1024)", 10)
stats_md("wgan", f"{tag}_wgan_kernel_stats.md", "python3 scripts/bench_wgan.py --steps 10 --warmup 2 (14 critic updates + 14 generator updates executed: 2 capture warm-ups, 2 warm-ups, 10 timed of each kind; a \"step\" in the table is the average update)", 28)

# DCGAN / CounteRGAN: the dispatches of ONE steady-state step, in order (scripts/step_census.py: between the step-ending Adam launches) —
# unlike the --stats table divided by the step count this leaves out set-up dispatches
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from step_census import census  # noqa: E402
for sub, dest, how in (("dcgan", f"{tag}_dcgan_launches.json", "python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-calib --no-secondary"),
                       ("countergan", f"{tag}_countergan_launches.json", "python3 scripts/bench_countergan.py --steps 5 --warmup 2 --no-cpu-baseline")):
    tr = find(sub, "*kernel_trace.csv")
    if tr:
        try:
            c = census(tr, "adam_kernel", 2, 1)
            c["how"] = f"dispatches after the previous step's second adam_kernel launch up to and including this step's, last complete step of rocprofv3 --kernel-trace -- {how}"
            json.dump(c, open(os.path.join(out, dest), "w"), indent=1)
        except SystemExit as e:
            print("census", sub, e)

# house: launches per step = dispatches between two consecutive house_draws_kernel launches in the steady state
trace = find("house", "*kernel_trace.csv")
if trace:
    rows = sorted(csv.DictReader(open(trace)), key=lambda r_: int(r_["Start_Timestamp"]))
    marks = [i for i, r_ in enumerate(rows) if "house_draws_kernel" in r_["Kernel_Name"]]
    if len(marks) > 12:
        a, b = marks[-11], marks[-1]
        lps = (b - a) / 10.0
        period = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 10.0 / 1e3
        busy = sum(int(r_["End_Timestamp"]) - int(r_["Start_Timestamp"]) for r_ in rows[a:b]) / 10.0 / 1e3
        nat = sum(1 for r_ in rows[a:b] if "at::native" in r_["Kernel_Name"]) / 10.0
        queues = sorted({r_["Queue_Id"] for r_ in rows[a:b]})
        per_name = collections.Counter(short(r_["Kernel_Name"]) for r_ in rows[a:b])
        json.dump({"launches_per_step": lps, "step_period_us_under_profiler": period, "sum_of_kernel_durations_us": busy,
                   "aten_kernels_per_step": nat, "hw_queues_used": len(queues),
                   "copy_kernels_per_step": sum(v for k, v in per_name.items() if "copyBuffer" in k or "fillBuffer" in k) / 10.0,
                   "dispatches_per_step_by_kernel": {k: v / 10.0 for k, v in sorted(per_name.items())},
                   "how": "dispatches between consecutive house_draws_kernel launches (one per step), mean of the last 10 steps of "
                          "rocprofv3 --kernel-trace -- python3 scripts/bench_house.py --steps 50 --warmup 10"},
                  open(os.path.join(out, f"{tag}_house_launches.json"), "w"), indent=1)
    stats_md("house", f"{tag}_house_kernel_stats.md", "python3 scripts/bench_house.py --steps 50 --warmup 10 (batch 4096, single-stream HIP-graph replay of the scheduled step)", 63)

# PMC traffic (the DCGAN bench; the counteRGAN and WGAN benches the same way when their passes were collected)
def pmc_summary(fetch_sub, write_sub, dest, cmd):
    fe_p, wr_p = find(fetch_sub, "*counter_collection.csv"), find(write_sub, "*counter_collection.csv")
    if not (fe_p and wr_p):
        return

    def agg(path, counter):
        d = collections.defaultdict(lambda: [0, 0.0])
        for r_ in csv.DictReader(open(path)):
            if r_["Counter_Name"] == counter:
                k = short(r_["Kernel_Name"])
                d[k][0] += 1
                d[k][1] += float(r_["Counter_Value"])
        return d
    fe, wr = agg(fe_p, "FETCH_SIZE"), agg(wr_p, "WRITE_SIZE")
    res = {}
    for k in fe:
        n = fe[k][0]
        res[k] = {"launches": n, "fetch_bytes_per_launch": 2 * 1024 * fe[k][1] / n,
                  "write_bytes_per_launch": 1024 * wr.get(k, [1, 0.0])[1] / max(wr.get(k, [1, 0.0])[0], 1)}
    fam = [k for k in res if k.startswith("conv_")]
    n = sum(res[k]["launches"] for k in fam)
    res["_igemm_family"] = {
        "launches": n,
        "hbm_bytes_per_launch": sum(res[k]["launches"] * (res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]) for k in fam) / n,
        "note": f"FETCH_SIZE doubled (gfx950 correction), separate --pmc passes for FETCH_SIZE and WRITE_SIZE; {cmd}"}
    json.dump(res, open(os.path.join(out, dest), "w"), indent=1, sort_keys=True)


pmc_summary("pmc_fetch", "pmc_write", f"{tag}_pmc_traffic.json", "bench.py --steps 2 --warmup 1")
pmc_summary("pmc_fetch_countergan", "pmc_write_countergan", f"{tag}_pmc_traffic_countergan.json", "scripts/bench_countergan.py --steps 2 --warmup 1")
pmc_summary("pmc_fetch_wgan", "pmc_write_wgan", f"{tag}_pmc_traffic_wgan.json", "scripts/bench_wgan.py --steps 2 --warmup 1")

for name in ("bench_line", "bench_line_dp1", "countergan_line", "wgan_line", "house_line"):
    p = os.path.join(src, name + ".json")
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            open(os.path.join(out, f"{tag}_{name}.json"), "w").write(lines[-1])
print("wrote", sorted(f for f in os.listdir(out) if f.startswith(tag)))
