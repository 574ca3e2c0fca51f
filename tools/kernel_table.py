#!/usr/bin/env python3
"""Per-kernel resource table of the gfx950 build: VGPRs, SGPRs, spills, occupancy, LDS (from
-Rpass-analysis=kernel-resource-usage) and the atomic instructions each kernel contains (counted in the
generated ISA). The ISA listing itself is NOT kept in the tree (84 K lines of compiler output that goes
stale); this table is what DESIGN.md's claims about native f64 atomics / no CAS loops refer to.

    python tools/kernel_table.py > profiles/rNN_kernel_resources.md      (needs hipcc; no GPU)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dynamicppr_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout
    return [re.sub(r"\(.*", "", x).replace("void ", "").replace("dppr::", "") for x in out.splitlines()]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "engine.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, "dppr_engine.hip"),
                            "-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
        if r.returncode:
            sys.exit(r.stderr[-2000:])
        res, cur = {}, None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: +(Function Name|[A-Za-z /\[\]]+): +(\S+)", line)
            if not m:
                continue
            key, val = m.group(1).strip(), m.group(2)
            if key == "Function Name":
                cur = val
                res[cur] = {}
            elif cur:
                res[cur][key] = val
        text = open(asm).read()
    bodies = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\s*\.amdhsa_kernel \1", text, flags=re.S | re.M):
        bodies[m.group(1)] = m.group(2)   # label ... kernel descriptor: all of the function's code (several s_endpgm)
    names = [n for n in res if "dppr" in n]
    pretty = demangle(names)
    print("# Kernel resources, gfx950 build (`tools/kernel_table.py`; flags: %s)\n" % " ".join(FLAGS))
    print("| kernel | VGPR | SGPR | VGPR spill | SGPR spill | waves/SIMD | LDS B | global f64 atomic add | global atomic swap/int | cmpswap | LDS f64 add |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for n, p in sorted(zip(names, pretty), key=lambda t: t[1]):
        d, b = res[n], bodies.get(n, "")
        cnt = lambda pat: len(re.findall(pat, b))  # noqa: E731
        print(f"| `{p}` | {d.get('VGPRs', '?')} | {d.get('SGPRs', d.get('TotalSGPRs', '?'))} | {d.get('VGPRs Spill', '?')} | "
              f"{d.get('SGPRs Spill', '?')} | {d.get('Occupancy [waves/SIMD]', '?')} | {d.get('LDS Size [bytes/block]', '?')} | "
              f"{cnt(r'global_atomic_add_f64')} | {cnt(r'global_atomic_(swap|add_u32|add_u64|or|umin|smin|add )')} | "
              f"{cnt(r'cmpswap')} | {cnt(r'ds_add(_rtn)?_f64')} |")


if __name__ == "__main__":
    main()
