#!/usr/bin/env python3
"""Per-kernel register / scratch table of one HIP source, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python scripts/kernel_resources.py decode_tonal_langauge_amd/csrc/tonal_wino63.hip [extra hipcc flags] [--filter nt_kernel]

Prints a markdown table (kernel, SGPRs, VGPRs, AGPRs, scratch bytes per lane, SGPR / VGPR spills, LDS bytes, waves per SIMD);
profiles/rNN_kernel_resources.md is this output for the files of the NT63 family."""
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    src, extra = args[0], args[1:]
    if not extra and src.endswith(("tonal_wino63.hip", "tonal_wino43_tn.hip")):
        # the Makefile builds these two files without packed fp32 arithmetic: with the default target features the allocator's
        # result is a different one (MASKY: 63 spilled registers instead of none)
        extra = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
           "-c", src, "-o", "/dev/null"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = demangle([r["name"] for r in rows])
    print("| kernel | SGPRs | VGPRs | AGPRs | scratch B/lane | SGPR spills | VGPR spills | LDS B | waves/SIMD |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void ", "").replace("tl::", "")
        if flt and flt not in n:
            continue
        print(f"| `{n}` | {r.get('TotalSGPRs')} | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('ScratchSize')} | "
              f"{r.get('SGPRs Spill')} | {r.get('VGPRs Spill')} | {r.get('LDS Size')} | {r.get('Occupancy')} |")


if __name__ == "__main__":
    main()
