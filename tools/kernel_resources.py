#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of a translation unit, and instruction-class counts of one kernel,
from the gfx950 assembly hipcc emits (no GPU needed).
    tools/kernel_resources.py render_bwd [substring of a kernel name to count instructions for] [-DNAME ...]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = sys.argv[1]
pick = next((a for a in sys.argv[2:] if not a.startswith("-D")), None)
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
       *defs, "-S", "--cuda-device-only", os.path.join(ROOT, "diffus_amd", "csrc", unit + ".hip"), "-o", "-"]
asm = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def demangle(n):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
            if out:
                return out
        except Exception:
            pass
    return n


meta = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", asm)
lds = dict(re.findall(r"\.amdhsa_kernel (\S+)\n(?:.*\n)*?\s+\.amdhsa_group_segment_fixed_size (\d+)", asm))
print("%-110s %5s %5s %7s %7s" % ("kernel", "vgpr", "sgpr", "lds", "scratch"))
for name, scratch, sgpr, vgpr in meta:
    d = demangle(name).replace("(anonymous namespace)::", "").replace("void ", "")
    d = re.sub(r"\(diffus::Args\)|\(.*\)$", "", d)
    print("%-110s %5s %5s %7s %7s" % (d[:110], vgpr, sgpr, lds.get(name, "?"), scratch))
if pick:
    for name, *_ in meta:
        d = demangle(name)
        if pick.replace(" ", "") in d.replace(" ", "") or pick in name:
            body = asm[asm.index(name + ":"):]
            body = body[:body.index("s_endpgm")]
            ops = collections.Counter()
            for line in body.splitlines():
                m = re.match(r"\s+([a-z_0-9]+)\s", line)
                if m:
                    op = m.group(1)
                    cls = ("dpp" if "dpp" in line else "ds_bpermute" if op.startswith("ds_bpermute") else "lds" if op.startswith("ds_") else
                           "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "valu" if op.startswith("v_") else
                           "waitcnt" if op.startswith("s_waitcnt") else "nop" if op.startswith("s_nop") else "salu" if op.startswith("s_") else "other")
                    ops[cls] += 1
                    if op in ("v_readlane_b32", "v_readfirstlane_b32"):
                        ops["readlane"] += 1
            print("\nstatic instruction counts of", d[:150])
            print("  ", dict(ops))
            break
