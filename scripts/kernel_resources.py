#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a .hip source, from the code object metadata hipcc emits for gfx950 (no GPU
needed):  python scripts/kernel_resources.py promptable-counterfactual-gan_amd/csrc/conv_igemm.hip [more.hip ...]
Prints vgpr / agpr / sgpr counts, spilled VGPRs, scratch bytes (private_segment_fixed_size) and static LDS per kernel — the
numbers VERDICT r02 asked to be zero for the conv_* kernels (spills)."""
import os
import re
import subprocess
import sys
import tempfile


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except FileNotFoundError:
        return names


def main():
    for src in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", asm, os.path.abspath(src)],
                               capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src)))
            if r.returncode:
                sys.exit(r.stderr)
            s = open(asm).read()
        rows = []
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", s, re.S):
            blk = m.group(0)
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            rows.append((re.search(r"\.name:\s+(\S+)", blk).group(1), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"),
                         g("private_segment_fixed_size"), g("group_segment_fixed_size")))
        names = demangle([r[0] for r in rows])
        print(f"# {src}")
        for nm, r in zip(names, rows):
            nm = re.sub(r"pcg::\(anonymous namespace\)::", "", nm)
            nm = re.sub(r"pcg::TileCfg<(\d+), (\d+), \d+, \d+, (\w+), (\d+), (\d+)>", r"T<\1x\2,swz=\3,w\4,pf\5>", nm)
            print(f"{nm[:120]:120s} vgpr {r[1]:3d} agpr {r[2]:3d} sgpr {r[3]:3d} spill {r[4]:3d} scratch {r[5]:4d} lds {r[6]}")


if __name__ == "__main__":
    main()
