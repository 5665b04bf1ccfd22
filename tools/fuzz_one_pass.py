#!/usr/bin/env python3
"""Diagnostic: random shapes, poses and volumes -- the one-pass training step (CapturedStep) against the two-call
autograd path (render_poses + torch loss) on the GPU.  usage: tools/fuzz_one_pass.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import diffus_amd as da

cases = int(sys.argv[1]) if (__name__ == "__main__" and len(sys.argv) > 1) else 100
rng = np.random.default_rng(int(sys.argv[2]) if (__name__ == "__main__" and len(sys.argv) > 2) else 0)
worst = {"frame": 0.0, "loss": 0.0, "gvol": 0.0, "gsrc": 0.0, "gdirs": 0.0}
ill = 0
def gen_case(rng):
    dims = tuple(int(x) for x in rng.integers(5, 70, 3))
    if rng.random() < 0.15:
        dims = dims[:2] + (1,)
    P, R = int(rng.integers(1, 5)), int(rng.integers(2, 40))
    S = int(rng.choice([2, 3, 17, 64, 65, 130, 300, 513, 700, 1024, 1025, 1400]))
    start = int(rng.integers(0, max(1, min(S - 2, 40)))) if (S > 3 and rng.random() < 0.5) else 0
    sampler = "trilinear" if rng.random() < 0.7 else "nearest"
    layout = str(rng.choice(["paired", "bricked", "canonical"]))
    alpha = float(rng.choice([1e-4, 1e-2, 0.5]))
    vol = (1.5e6 + 3e5 * rng.standard_normal(dims)).astype(np.float32)
    if rng.random() < 0.2:
        vol[tuple(rng.integers(0, d) for d in dims)] = 0.0
    centre = np.array(dims, np.float64) / 2
    src = centre + rng.standard_normal((P, 3)) * np.array(dims) * (0.8 if rng.random() < 0.5 else 0.3)
    dirs = rng.standard_normal((P, R, 3))
    if rng.random() < 0.6:
        dirs[..., 2] = 0.0                                  # planar fans, like every fan of the reference
    dirs /= np.maximum(np.linalg.norm(dirs, axis=-1, keepdims=True), 1e-9)
    dirs *= rng.choice([1.0, 0.5, 40.0 / S])
    f64 = rng.random() < 0.25
    sdt = np.float64 if f64 else np.float32
    tgt = (0.05 * rng.standard_normal((P, R, S - start))).astype(np.float32)
    scale = float(rng.choice([1.0, 0.37]))
    return dict(dims=dims, P=P, R=R, S=S, start=start, sampler=sampler, layout=layout, alpha=alpha, vol=vol,
                src=src.astype(sdt), dirs=dirs.astype(sdt), f64=f64, tgt=tgt, scale=scale)


if __name__ != "__main__":
    cases = 0
for c in range(cases):
    k = gen_case(rng)
    dims, P, R, S, start, sampler, layout, alpha, vol, tgt, scale, f64 = (k[x] for x in (
        "dims", "P", "R", "S", "start", "sampler", "layout", "alpha", "vol", "tgt", "scale", "f64"))
    src, dirs, sdt = k["src"], k["dirs"], k["src"].dtype
    v = torch.from_numpy(vol).cuda()
    s = torch.from_numpy(src).cuda()
    d = torch.from_numpy(dirs).cuda()
    t = torch.from_numpy(tgt).cuda()
    one = da.CapturedStep(v, s, d, S, alpha, sampler, start=start, layout=layout, target=t, loss_scale=scale)
    one.step()
    v2 = v.clone().requires_grad_(True); s2 = s.clone().requires_grad_(True); d2 = d.clone().requires_grad_(True)
    f = da.render_poses(v2, s2, d2, S, alpha, start=start, sampler=sampler, layout=layout)
    loss = scale * ((f - t) ** 2).sum(dim=(1, 2))
    loss.sum().backward()
    torch.cuda.synchronize()

    def rel(a, b):
        m = float(b.abs().max())
        return float((a - b).abs().max()) / m if m > 0 else float((a - b).abs().max())

    e = {"frame": rel(one.frame, f.detach()), "loss": rel(one.loss, loss.detach()), "gvol": rel(one.gvol, v2.grad),
         "gsrc": rel(one.gsrc.double(), s2.grad.double()), "gdirs": rel(one.gdirs.double(), d2.grad.double())}
    # a gradient that is nothing but the float32 residue of cancelling terms (every sample clamped onto a border voxel,
    # or an upstream gradient attenuated to nothing: max |g| < 1e-10 where single terms are ~1e-7) is not compared
    if float(v2.grad.abs().max()) < 1e-10:
        e["gvol"] = 0.0
    bad = (not all(np.isfinite(x) for x in e.values())) or e["frame"] > 5e-5 or e["loss"] > 1e-4 or e["gvol"] > 2e-3 \
        or e["gsrc"] > 2e-3 or e["gdirs"] > 2e-3
    for k in worst:
        if np.isfinite(e[k]):
            worst[k] = max(worst[k], e[k])
    if c % 500 == 499:
        print("progress", c + 1, {k: "%.1e" % x for k, x in worst.items()}, flush=True)
    if bad:
        # An echo is the ratio (P_n)01 / (P_n)11 of a running matrix product; where the denominator nearly cancels (|echo| > 1: rays
        # through white noise like this tool's volumes) float32 loses digits.  The two-call path's FORWARD kernel evaluates such a ray
        # again in float64, the one-pass step only on request (CapturedStep(repair_frames=True)): there the two are expected to
        # differ, and the case is re-run with the repair on before it counts as a mismatch.
        emax = float((f.detach().double() * torch.exp(alpha * torch.arange(f.shape[-1], device=f.device, dtype=torch.float64))).abs().max())   # (float64: exp(0.5 * 1023) is beyond float32)
        kind = "MISMATCH"
        if emax > 1.0:
            rep = da.CapturedStep(v, s, d, S, alpha, sampler, start=start, layout=layout, target=t, loss_scale=scale, repair_frames=True)
            rep.step()
            torch.cuda.synchronize()
            e["frame_repaired"] = rel(rep.frame, f.detach())
            e["loss_repaired"] = rel(rep.loss, loss.detach())
            kind = "ILL-CONDITIONED" if (e["frame_repaired"] <= 5e-5 and e["loss_repaired"] <= 1e-4) else "MISMATCH"
            ill += kind == "ILL-CONDITIONED"
        print(kind, "case", c, dims, P, R, S, start, sampler, layout, alpha, "f64" if f64 else "f32", "max |echo| %.1f" % emax, e, flush=True)
if cases:
    print("cases", cases, "worst relative differences", {k: "%.1e" % x for k, x in worst.items()}, "ill-conditioned (equal once repaired):", ill)
