#!/usr/bin/env python3
"""Issue-cost model of one kernel: its gfx950 disassembly (hipcc -S, no GPU needed) weighted with the per-instruction
issue costs MEASURED by tools/valu_issue_bench.hip (profiles/r04_valu_issue_bench.txt; cycles per wave-instruction per SIMD
with >= 4 waves per SIMD).  Prints the opcode histogram sorted by cost and the kernel's lower bound in SIMD cycles per
wave.  The render kernels are straight-line code with a few wave-uniform branches; -D defines pick the executed copy
(e.g. -DDIFFUS_COUNT_PLANAR=1 -DDIFFUS_COUNT_MSE=2), so the static count is the dynamic one.

    tools/issue_model.py render_bwd 'render_bwd_kernel<8, 1, 2, true, 4, 0' [-DNAME=VALUE ...] [--json]
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# cycles per wave-instruction per SIMD at >= 4 waves per SIMD (tools/valu_issue_bench.hip on MI355X, round 4)
FULL, HALF, TRANS, F64 = 2.25, 4.25, 8.2, 4.65
FULL_OPS = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_mac_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
            "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32",
            "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_mul_legacy_f32", "v_madak_f32", "v_madmk_f32", "v_fmaak_f32", "v_fmamk_f32",
            "v_ashrrev_i32", "v_nop"}
TRANS_OPS = {"v_rcp_f32", "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def cost_of(op: str, line: str) -> float:
    if not op.startswith("v_"):
        return 0.0
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if "dpp" in op or " row_" in line or " wave_" in line or "quad_perm" in line:
        return HALF
    if base in TRANS_OPS:
        return TRANS
    if base.endswith("_f64") or "_f64_" in base:
        return F64
    if base.startswith("v_pk_"):
        return 4.55
    if base in FULL_OPS:
        return FULL
    return HALF          # everything else measured: compares, selects, min/max, cvt, floor, ldexp, 24-bit multiplies, 3-operand integer ops


def kernel_body(asm: str, pick: str):
    names = re.findall(r"^(_Z\S+):", asm, re.M)
    filt = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for name, dem in zip(names, filt):
        if pick.replace(" ", "") in dem.replace(" ", ""):
            body = asm[asm.index(name + ":"):]
            return dem, body[:body.index("s_endpgm")]
    raise SystemExit(f"no kernel matching {pick!r}")


def emit(path):
    """profiles/<round>_issue_model.json: the executed copies of the step's two big kernels (bench.py reads it)."""
    recs = []
    for short, unit, pick, defs in (
            ("render_bwd_kernel", "render_bwd", "render_bwd_kernel<8, 1, 2, true, 4, 0, false, 1, true>", ["-DDIFFUS_COUNT_PLANAR=1", "-DDIFFUS_COUNT_FAST_ONLY"]),
            ("scatter_patch_kernel", "scatter", "scatter_patch_kernel<1, 1, 0>", ["-DDIFFUS_SC_PLANAR_ONLY"])):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), unit, pick, *defs, "--json"], check=True,
                             capture_output=True, text=True).stdout
        rec = json.loads(out)
        rec["short"] = short
        recs.append(rec)
    json.dump({"costs": {"full_rate": FULL, "half_rate": HALF, "transcendental": TRANS, "f64": F64, "packed_f32": 4.55,
                         "source": "profiles/r04_valu_issue_bench.txt (tools/valu_issue_bench.hip on MI355X, >= 4 waves per SIMD)"},
               "note": "static VALU instruction mix of the executed copy of each kernel (config 3: planar fans, f32 poses, paired "
                       "volume) priced with the measured per-instruction issue costs",
               "kernels": recs}, open(path, "w"), indent=1)
    for r in recs:
        print(r["short"], r["static_counts"], "%.0f cycles per wave, %.2f per VALU instruction" % (r["valu_issue_cycles_per_wave"], r["mean_cycles_per_valu"]))


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--emit":
        return emit(sys.argv[2])
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    unit, pick = args[0], args[1]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           *defs, "-S", "--cuda-device-only", os.path.join(ROOT, "diffus_amd", "csrc", unit + ".hip"), "-o", "-"]
    asm = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    dem, body = kernel_body(asm, pick)
    hist = collections.Counter()
    cost = collections.Counter()
    kinds = collections.Counter()
    phase, phase_cost, phase_n = "start", collections.OrderedDict(), collections.Counter()
    for line in body.splitlines():
        pm = re.search(r"; DIFFUS_PHASE (\S+)", line)
        if pm:
            phase = "after mark " + pm.group(1)
            continue
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", line)
        if not m:
            continue
        if m.group(1).startswith("v_"):
            phase_cost[phase] = phase_cost.get(phase, 0.0) + cost_of(m.group(1), line)
            phase_n[phase] += 1
        op = m.group(1)
        if op.startswith("v_"):
            key = op + ("(dpp)" if ("row_" in line or "wave_" in line or "quad_perm" in line) and "dpp" not in op else "")
            hist[key] += 1
            cost[key] += cost_of(op, line)
            kinds["valu"] += 1
        elif op.startswith("s_"):
            kinds["salu" if not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch")) else "sctl"] += 1
        elif op.startswith("ds_"):
            kinds["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            kinds["vmem"] += 1
    total = sum(cost.values())
    n_valu = kinds["valu"]
    out = {"kernel": dem, "defines": defs, "static_counts": dict(kinds), "valu_issue_cycles_per_wave": total,
           "mean_cycles_per_valu": total / max(n_valu, 1),
           "issue_slot_cycles_per_wave": FULL * (n_valu + kinds["salu"] + kinds["lds"] + kinds["vmem"])}
    if "--json" in sys.argv:
        out["top"] = [[k, hist[k], round(cost[k], 1)] for k, _ in cost.most_common(25)]
        print(json.dumps(out))
        return
    print(dem[:160])
    print("static:", dict(kinds))
    print(f"VALU issue cost {total:.0f} cycles per wave ({total / max(n_valu, 1):.2f} per instruction); every instruction at one issue slot "
          f"of {FULL}: {out['issue_slot_cycles_per_wave']:.0f}")
    for k, c in cost.most_common(40):
        print(f"  {k:28s} {hist[k]:5d}  {c:8.1f}")
    if len(phase_cost) > 1:
        print("phases (static order of the marks in the assembly; branches make some of them alternatives):")
        for k, c in phase_cost.items():
            print(f"  {k:18s} {phase_n[k]:5d} VALU {c:8.1f} cycles")


if __name__ == "__main__":
    main()
