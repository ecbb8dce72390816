#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/prof*/, gpurun_out/pmc_*/) into the small, tracked summaries under profiles/.

  python scripts/summarize_profiles.py <round-tag> <kernel_stats.csv> <steps> [<pmc_fetch.csv> <pmc_write.csv>]
FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide coalesced reads: doubled here, as MI355X_MICROARCH.md
prescribes; WRITE_SIZE is exact.  Both are reported by rocprofv3 in KiB.
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def short(name):
    name = name.replace("pcg::(anonymous namespace)::", "").replace("void ", "")
    name = name.replace("pcg::TileCfg<128, 128, 2, 2>", "128x128").replace("pcg::TileCfg<128, 64, 2, 2>", "128x64")
    name = name.replace("pcg::TileCfg<64, 128, 1, 4>", "64x128")
    return name.split("(")[0]


rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(out_dir, f"{tag}_kernel_stats.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats, bench.py ({steps} steps profiled incl. warm-up), round {tag}\n\n")
    f.write("| kernel | launches/step | avg µs | ms/step | % of GPU time |\n|---|---|---|---|---|\n")
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.0005:
            continue
        f.write(f"| `{short(r['Name'])}` | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
                f"{t / 1e6 / steps:.3f} | {100 * t / tot:.1f} |\n")
    f.write(f"\nGPU busy time: {tot / 1e6 / steps:.3f} ms/step\n")

if len(sys.argv) > 5:
    def agg(path, counter):
        d = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                d[k][0] += 1
                d[k][1] += float(r["Counter_Value"])
        return d
    fe, wr = agg(sys.argv[4], "FETCH_SIZE"), agg(sys.argv[5], "WRITE_SIZE")
    res = {}
    for k in fe:
        n = fe[k][0]
        res[k] = {"launches": n, "fetch_bytes_per_launch": 2 * 1024 * fe[k][1] / n,
                  "write_bytes_per_launch": 1024 * wr.get(k, [1, 0.0])[1] / max(wr.get(k, [1, 0.0])[0], 1)}
    fam = [k for k in res if k.startswith("conv_")]
    n = sum(res[k]["launches"] for k in fam)
    res["_igemm_family"] = {
        "launches": n,
        "hbm_bytes_per_launch": sum(res[k]["launches"] * (res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"])
                                    for k in fam) / n,
        "note": "FETCH_SIZE doubled (gfx950 correction), separate --pmc passes for FETCH_SIZE and WRITE_SIZE"}
    json.dump(res, open(os.path.join(out_dir, f"{tag}_pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("wrote", os.listdir(out_dir))
