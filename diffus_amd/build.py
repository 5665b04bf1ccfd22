"""Builds diffus_amd/libdiffus_hip.so (gfx950) with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/diffus_hip.h).  The eight translation units under csrc/ are
compiled in parallel and linked."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
UNITS = ("render_fwd", "render_bwd", "scatter", "splat", "artifacts", "impedance", "ssim", "pose")
HEADERS = (os.path.join(CSRC, "diffus_device.hpp"), os.path.join(CSRC, "diffus_host.hpp"),
           os.path.join(ROOT, "include", "diffus_hip.h"))
OBJDIR = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libdiffus_hip.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


def needs_build() -> bool:
    return _stale(OUT, [os.path.join(CSRC, u + ".hip") for u in UNITS] + list(HEADERS))


def build(force: bool = False, verbose: bool = False, defines=(), out: str = OUT) -> str:
    """Compile (only what changed) and link.  `defines` / `out` serve the diagnostic builds in tools/."""
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    tag = "" if not defines else "_" + "_".join(d.replace("=", "-") for d in defines)
    dflags = ["-D" + d for d in defines]

    def compile_unit(u):
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(OBJDIR, u + tag + ".o")
        if force or _stale(obj, [src, *HEADERS]):
            cmd = [cc, *FLAGS, *dflags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        return obj

    if force or defines or _stale(out, [os.path.join(CSRC, u + ".hip") for u in UNITS] + list(HEADERS)):
        with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
            objs = list(ex.map(compile_unit, UNITS))
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[2:] for a in sys.argv[1:] if a.startswith("-o")]
    print(build(force="--force" in sys.argv, verbose=True, defines=defs, out=outs[0] if outs else OUT))
