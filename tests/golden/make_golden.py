#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--big]

The script imports /root/reference/src/renderer.py unmodified and records plain
input/output arrays; it contains none of the reference's source and copies none
into the repo.  It exits cleanly when /root/reference is absent (GPU box).
Golden cases follow SURVEY.md §8(c) G1-G9 (+ G10 with --big: the 256x512
config-2 shape, ~2.5 minutes of dense solves).

`generate_cone_directions` lives in src/cone.py, whose module-level imports
(nibabel, cv2, ...) are not installed here (ordinary ModuleNotFoundError), so
that one function is pulled out of the file with `ast` at run time and executed
as is -- still the reference's own code, executed where it lies.
"""
from __future__ import annotations

import argparse
import ast
import contextlib
import io
import json
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also generate G10 (256 rays x 512 steps, ~150 s) and G19 (~15 min)")
    ap.add_argument("--only", default="", help="comma list of cases to (re)generate")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "src")):
        print("reference not present; nothing to do")
        return 0

    import numpy as np
    import torch

    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    import src.renderer as ref  # the reference, unmodified
    from diffus_amd.phantom import phantom, pose_ring

    torch.set_grad_enabled(False)
    only = set(filter(None, args.only.split(",")))

    def want(name):
        return not only or name in only

    def save(name, **arrs):
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrs)
        print(f"wrote {name}.npz ({os.path.getsize(path)/1024:.1f} KiB)")

    @contextlib.contextmanager
    def quiet():
        with contextlib.redirect_stdout(io.StringIO()):
            yield

    def ref_frame(vol, source, directions, S, alpha, start):
        """plot_beam_frame of the reference (non-grad inputs, artifacts off)."""
        rr = ref.UltrasoundRenderer(num_samples=S, attenuation_coeff=alpha)
        with quiet():
            x, y, z, f = rr.plot_beam_frame(volume=vol, source=source, directions=directions,
                                            plot=False, artifacts=False, start=start)
        import matplotlib.pyplot as plt
        plt.close("all")
        return x.numpy(), y.numpy(), z.numpy(), f.numpy()

    def ref_cone(direction, opening, n):
        tree = ast.parse(open(os.path.join(REF, "src", "cone.py")).read())
        fn = [n_ for n_ in tree.body if isinstance(n_, ast.FunctionDef) and n_.name == "generate_cone_directions"][0]
        ns = {"np": np, "torch": torch}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), "cone.py", "exec"), ns)
        return ns["generate_cone_directions"](direction, opening, n)

    # ---- G1: Z=(1,2,1.5) --------------------------------------------------------
    if want("g1"):
        Z = torch.tensor([[1.0, 2.0, 1.5]])
        r = ref.UltrasoundRenderer.compute_reflection_coeff(Z[:, :-1], Z[:, 1:])
        w = ref.prop_single_ray(r)
        e, _ = ref.compute_echo_traces(r)
        save("g1_three_layer", Z=Z.numpy(), r=r.numpy(), w=w.numpy(), echo=e.numpy())

    # ---- G2: 5x10 tumour phantom literal of `[DEMO] Modeling Choices` cell 6 -----
    if want("g2"):
        nb = json.load(open(os.path.join(REF, "notebooks", "[DEMO] Modeling Choices.ipynb")))
        src_lines = "".join(nb["cells"][6]["source"])
        lit = src_lines[src_lines.index("torch.tensor(") + len("torch.tensor("):]
        depth, end = 0, 0
        for i, ch in enumerate(lit):
            depth += ch == "["
            depth -= ch == "]"
            if depth == 0 and ch == "]":
                end = i + 1
                break
        Zp = torch.tensor(ast.literal_eval(lit[:end].replace("\n", " ")))
        r = ref.UltrasoundRenderer.compute_reflection_coeff(Zp[:, 1:], Zp[:, :-1])  # (sic) swapped, as in the notebook
        e, _ = ref.compute_echo_traces(r)
        save("g2_modeling_choices_phantom", Z=Zp.numpy(), r=r.numpy(), echo=e.numpy())

    # ---- G3: zero impedance -> NaN r -> echoes zeroed from there on --------------
    if want("g3"):
        Z = torch.tensor([[1.0, 1.0, 0.0, 0.0, 1.0, 1.0]])
        r = ref.UltrasoundRenderer.compute_reflection_coeff(Z[:, :-1], Z[:, 1:])
        e, _ = ref.compute_echo_traces(r)
        save("g3_nan", Z=Z.numpy(), r=r.numpy(), echo=e.numpy())

    # ---- G4: random reflection series, B=8, N=255, fp32 and fp64 -----------------
    if want("g4"):
        g = torch.Generator().manual_seed(1234)
        Z = 1.6e6 + 5e4 * torch.randn(8, 256, generator=g, dtype=torch.float64)
        Z[1, 100:104] = 6.4e6                 # bone plate
        Z[2, 200:] = 400.0                    # into air
        Z[3, 50:120] = Z[3, 50:51]            # repeated-voxel run (r = 0)
        Z[4] = 1.6e6 + 2e3 * torch.randn(256, generator=g, dtype=torch.float64)  # weak scatterers
        Z[5, ::2] = 400.0                     # adversarial: alternating air / tissue
        Z[6] = torch.linspace(1.4e6, 1.8e6, 256, dtype=torch.float64)
        Z32 = Z.float()
        r32 = ref.UltrasoundRenderer.compute_reflection_coeff(Z32[:, :-1], Z32[:, 1:])
        e32, _ = ref.compute_echo_traces(r32)
        r64 = r32.double()
        e64, _ = ref.compute_echo_traces(r64)
        save("g4_random_series", Z=Z32.numpy(), r=r32.numpy(), echo32=e32.numpy(), echo64=e64.numpy())

    # ---- G5: whole frames on small phantoms ------------------------------------
    if want("g5"):
        out = {}
        cases = []
        v32 = torch.from_numpy(phantom(32))
        v64 = torch.from_numpy(phantom(64))
        s64, d64 = pose_ring(64, 4, 16)
        s32, d32 = pose_ring(32, 4, 16)
        cases.append(("a", 64, torch.from_numpy(s64[0]), torch.from_numpy(d64[0]), 48, 1e-4, 0))
        cases.append(("b", 64, torch.from_numpy(s64[1]), torch.from_numpy(d64[1]), 48, 1e-4, 8))
        # NB a float `start` (src/renderer.py:237-238) cannot be exercised: the always-on
        # visualisation slices with it first and raises TypeError (src/renderer.py:774).
        cases.append(("c", 64, torch.from_numpy(s64[2]), torch.from_numpy(d64[2]), 48, 0.5, 12))
        cases.append(("d", 32, torch.from_numpy(s32[3]), torch.from_numpy(d32[3]), 48, 1e-3, 0))
        # source outside the volume: the clamp path (the norm in the demos)
        cases.append(("e", 64, torch.tensor([-5.0, -5.0, 32.0]),
                      ref_cone((1.0, 1.0), 0.6, 16), 48, 1e-4, 3))
        # float64 source (demos pass f64 apexes): the add is then done in f64
        cases.append(("f", 64, torch.tensor([40.123456789, 9.87654321, 30.5], dtype=torch.float64),
                      ref_cone((-0.2, 0.9), 0.85, 16), 48, 1e-4, 0))
        # float64 source AND directions
        cases.append(("g", 64, torch.tensor([20.1, 50.2, 31.7], dtype=torch.float64),
                      ref_cone((0.7, -0.7), 0.5, 16).double(), 40, 1e-2, 5))
        # integer source as in `[TEST] Testing plotbeamframe sub functions` cell 5
        cases.append(("h", 64, torch.tensor([10, 60, 30]), ref_cone((0.3, -0.95), 0.85, 16), 48, 1e-3, 0))
        # oblique (out-of-plane) rays + half-integer coordinates (round-half-even ties)
        dd = torch.tensor([[0.5, 0.5, 0.70710678], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.5, -0.5, 0.0]])
        cases.append(("i", 64, torch.tensor([10.5, 31.5, 2.5]), dd, 48, 1e-4, 0))
        for tag, n, s, d, S, alpha, start in cases:
            vol = v64 if n == 64 else v32
            x, y, z, f = ref_frame(vol, s, d, S, alpha, start)
            out[f"{tag}_n"] = np.int64(n); out[f"{tag}_S"] = np.int64(S)
            out[f"{tag}_alpha"] = np.float64(alpha); out[f"{tag}_start"] = np.float64(start)
            out[f"{tag}_start_is_float"] = np.bool_(isinstance(start, float))
            out[f"{tag}_source"] = s.numpy(); out[f"{tag}_directions"] = d.numpy()
            out[f"{tag}_x"] = x.astype(np.int16); out[f"{tag}_y"] = y.astype(np.int16); out[f"{tag}_z"] = z.astype(np.int16)
            out[f"{tag}_frame"] = f
        out["tags"] = np.array([c[0] for c in cases])
        save("g5_small_frames", **out)

    # ---- G6: config 1 (256^3 phantom, 64 rays x 256 steps) ----------------------
    if want("g6"):
        v = torch.from_numpy(phantom(256))
        s, d = pose_ring(256, 32, 64)
        x, y, z, f = ref_frame(v, torch.from_numpy(s[5]), torch.from_numpy(d[5]), 256, 1e-4, 0)
        save("g6_config1", pose=np.int64(5), P=np.int64(32), source=s[5], directions=d[5],
             x=x.astype(np.int16), y=y.astype(np.int16), z=z.astype(np.int16), frame=f)

    # ---- G7: d(sum frame^2)/d volume through the reference's sub-functions -------
    if want("g7"):
        torch.set_grad_enabled(True)
        v = torch.from_numpy(phantom(64)).clone().requires_grad_(True)
        s, d = pose_ring(64, 4, 16)
        S, alpha = 48, 1e-4
        src_t, dir_t = torch.from_numpy(s[0]), torch.from_numpy(d[0])
        steps = torch.arange(0, S, dtype=torch.float32).view(1, -1, 1)
        pts = src_t + steps * dir_t.unsqueeze(1)
        with quiet():
            x, y, z, imp = ref.custom_nearest_sampler(v, pts, visualize=False)
        r = ref.UltrasoundRenderer.compute_reflection_coeff(imp[:, :-1], imp[:, 1:])
        e, _ = ref.compute_echo_traces(r)
        frame = e * torch.exp(-alpha * torch.arange(e.shape[1]).float())[None, :]
        (frame ** 2).sum().backward()
        g = v.grad
        nz = g.flatten().nonzero().flatten()
        save("g7_volume_grad", n=np.int64(64), S=np.int64(S), alpha=np.float64(alpha), source=s[0], directions=d[0],
             frame=frame.detach().numpy(), grad_index=nz.numpy(), grad_value=g.flatten()[nz].numpy())
        torch.set_grad_enabled(False)

    # ---- G8: generate_cone_directions ------------------------------------------
    if want("g8"):
        out = {}
        cases = [((-0.3, -0.95), 0.85, 64), ((1.0, 0.0), 1.0471975511965976, 256), ((0.2, 0.7, 15.0), 0.3, 5),
                 ((-1.0, 0.4), 1.2, 200), ((0.0, -3.0), 2.0, 2),
                 (torch.tensor([-0.3, -0.95, 15.0]), 0.85, 64),          # float32 tensor, as the demos pass
                 (torch.tensor([3, -4]), 0.5, 7)]                        # integer tensor
        for j, (dvec, op, n) in enumerate(cases):
            out[f"c{j}_direction"] = np.array(dvec)            # dtype preserved: it changes the rounding
            out[f"c{j}_opening"] = np.float64(op)
            out[f"c{j}_n"] = np.int64(n)
            out[f"c{j}_out"] = ref_cone(dvec, op, n).numpy()
        out["ncases"] = np.int64(len(cases))
        # the 64-ray fan printed (4 decimals) by `[DEMO] Train MRI to Impedance MLP - GPU` cell 12
        nb = json.load(open(os.path.join(REF, "notebooks", "[DEMO] Train MRI to Impedance MLP - GPU.ipynb")))
        txt = "".join("".join(o.get("text", "")) for o in nb["cells"][12]["outputs"])
        body = txt[txt.index("tensor([[") + len("tensor("): txt.index("]])") + 2]
        out["nb_fan64"] = np.array(ast.literal_eval(body.replace("\n", " ")), dtype=np.float64)
        out["nb_fan64_opening_deg"] = np.float64(txt[txt.index("]])") + 3:].split()[0])
        save("g8_cone_directions", **out)

    # ---- G9: trilinear sampling (torch grid_sample) + reference echo chain -------
    if want("g9"):
        import torch.nn.functional as F
        v = torch.from_numpy(phantom(32))
        s, d = pose_ring(32, 4, 8)
        S, alpha = 40, 1e-3
        src_t, dir_t = torch.from_numpy(s[1]), torch.from_numpy(d[1]).clone()
        dir_t[:, 2] = 0.13  # tilt out of plane so all three lerps are exercised
        dir_t = dir_t / dir_t.norm(dim=1, keepdim=True)
        steps = torch.arange(0, S, dtype=torch.float32).view(1, -1, 1)
        pts = (src_t + steps * dir_t.unsqueeze(1)).double()
        n = 32
        grid = torch.stack([2 * pts[..., 2] / (n - 1) - 1, 2 * pts[..., 1] / (n - 1) - 1,
                            2 * pts[..., 0] / (n - 1) - 1], dim=-1).view(1, -1, S, 1, 3)
        imp = F.grid_sample(v.double()[None, None], grid, mode="bilinear", padding_mode="border",
                            align_corners=True).view(-1, S)
        r = ref.UltrasoundRenderer.compute_reflection_coeff(imp[:, :-1], imp[:, 1:])
        e, _ = ref.compute_echo_traces(r)
        frame = e * torch.exp(-alpha * torch.arange(e.shape[1]).double())[None, :]
        save("g9_trilinear", n=np.int64(n), S=np.int64(S), alpha=np.float64(alpha), source=src_t.numpy(),
             directions=dir_t.numpy(), imp=imp.numpy(), frame=frame.numpy())

    # ---- G11: differentiable_splat (+ rotate_around_apex), forward and autograd ----------
    if want("g11"):
        torch.set_grad_enabled(True)
        out = {}
        v = torch.from_numpy(phantom(64))
        s, d = pose_ring(64, 4, 32)
        cases = [("a", 0, 4, 64, 64, 1.5, None), ("b", 1, 0, 64, 64, 2.0, None), ("c", 2, 10, 48, 80, 1.0, None),
                 ("d", 3, 0, 64, 64, 2.0, "rot"), ("e", 0, 4, 64, 64, 0.7, "perm")]
        for tag, p, start, H, W, sigma, mode in cases:
            with torch.no_grad():
                x, y, z, f = (torch.from_numpy(a) for a in ref_frame(v, torch.from_numpy(s[p]), torch.from_numpy(d[p]), 64, 1e-3, start))
            if mode == "rot":       # float coordinates, as after rotate_around_apex in the REUBEN demos
                with quiet():
                    xr, yr = ref.rotate_around_apex(x.flatten().float(), y.flatten().float(), (32.0, 5.0), (float(-d[p][16][0]), float(-d[p][16][1])))
                out[f"{tag}_rot_x"] = xr.numpy(); out[f"{tag}_rot_y"] = yr.numpy()
                x, y = xr.reshape(x.shape), yr.reshape(y.shape)
            if mode == "perm":      # axes permuted as in `[DEMO] CT Render Lung` cell 17
                x, y, z = z, x, y
            fi = f.clone().requires_grad_(True)
            with quiet():
                o = ref.differentiable_splat(x, y, z, fi, H=H, W=W, sigma=sigma)
            g = torch.Generator().manual_seed(7)
            up = torch.randn(o.shape, generator=g)
            (o * up).sum().backward()
            out[f"{tag}_x"] = x.numpy(); out[f"{tag}_y"] = y.numpy(); out[f"{tag}_z"] = z.numpy()
            out[f"{tag}_f"] = f.numpy(); out[f"{tag}_H"] = np.int64(H); out[f"{tag}_W"] = np.int64(W)
            out[f"{tag}_sigma"] = np.float64(sigma); out[f"{tag}_out"] = o.detach().numpy()
            out[f"{tag}_up"] = up.numpy(); out[f"{tag}_grad"] = fi.grad.numpy()
        out["tags"] = np.array([c[0] for c in cases])
        save("g11_splat", **out)
        torch.set_grad_enabled(False)

    # ---- G12: probe-pose helpers of src/cone.py (executed from the file via ast, like G8) -------
    if want("g12"):
        tree = ast.parse(open(os.path.join(REF, "src", "cone.py")).read())
        names = ("voxel_to_world", "world_to_voxel", "compute_us_apex_and_direction", "cone_us_to_mri_world")
        fns = [n_ for n_ in tree.body if isinstance(n_, ast.FunctionDef) and n_.name in names]
        ns = {"np": np, "torch": torch}
        exec(compile(ast.Module(body=fns, type_ignores=[]), "cone.py", "exec"), ns)
        rng = np.random.default_rng(5)
        out = {}
        lines = [(-1.7, 300.0, 1.4, -60.0), (-0.6, 120.5, 0.9, 10.25), (-3.0, 50.0, 2.0, 75.0)]
        for j, (ml, bl, mr, br) in enumerate(lines):
            r = ns["compute_us_apex_and_direction"](ml, bl, mr, br)
            out[f"l{j}_in"] = np.array([ml, bl, mr, br]); out[f"l{j}_apex"] = np.array(r["apex"])
            out[f"l{j}_opening"] = np.float64(r["opening_angle"]); out[f"l{j}_dir"] = np.array(r["direction_vector"])
        def affine():
            A = np.eye(4); q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            A[:3, :3] = q * rng.uniform(0.4, 1.2, 3); A[:3, 3] = rng.normal(0, 40, 3)
            return A
        for j in range(3):
            Au, At = affine(), affine()
            apex = rng.uniform(0, 200, 3); d2 = rng.normal(size=2)
            a, dv = ns["cone_us_to_mri_world"](apex, d2, Au, At)
            out[f"a{j}_Au"] = Au; out[f"a{j}_At"] = At; out[f"a{j}_apex"] = apex; out[f"a{j}_d2"] = d2
            out[f"a{j}_apex_t1"] = a; out[f"a{j}_dir_t1"] = dv
            out[f"a{j}_v2w"] = ns["voxel_to_world"](apex, Au); out[f"a{j}_w2v"] = ns["world_to_voxel"](apex, At)
        out["nl"] = np.int64(len(lines)); out["na"] = np.int64(3)
        save("g12_pose_helpers", **out)

    # ---- G13: compute_gaussian_pulse / gaussian_pulse (`[DEMO] Modeling Choices` cells 8-12) -----
    if want("g13"):
        g4 = np.load(os.path.join(HERE, "g4_random_series.npz"))
        r = torch.from_numpy(g4["r"][:, :120].copy())
        out = {"r": r.numpy()}
        for j, (length, sigma) in enumerate([(10, 1), (20, 4), (7, 2)]):
            out[f"p{j}"] = np.array([length, sigma])
            out[f"pulse{j}"] = ref.gaussian_pulse(length, sigma)
            out[f"out{j}"] = ref.compute_gaussian_pulse(r, length=length, sigma=sigma).numpy()
        save("g13_gaussian_pulse", **out)

    # ---- G14: artifact chain of plot_beam_frame(artifacts=True) with a SEEDED NumPy RNG ----------
    if want("g14"):
        out = {}
        v = torch.from_numpy(phantom(64))
        s, d = pose_ring(64, 4, 24)
        cases = [("a", 0, 60, 0, 0.01, 0.15, 4.0, 5.0), ("b", 1, 64, 6, 0.1, 0.02, 2.0, 1.5), ("c", 2, 40, 0, 0.05, 0.05, 0.5, 3.0)]
        for tag, p, S, start, std_r, std_l, max_sigma, alpha in cases:
            x, y, z, f = ref_frame(v, torch.from_numpy(s[p]), torch.from_numpy(d[p]), S, 1e-3, start)
            f = torch.from_numpy(f)
            R, N = f.shape
            np.random.seed(1000 + p)
            a1 = ref.add_speckle_arcs_np(f.clone(), std_radial=std_r, std_local=std_l)
            a2 = ref.add_depth_dependent_lateral_blur_np(a1.clone(), max_sigma=max_sigma)
            a3 = ref.sharpen_np(a2.clone(), alpha=alpha)
            np.random.seed(1000 + p)          # the same draws, in the same order (:567-574)
            depth = np.linspace(0.0, 1.0, N)
            radial = np.random.normal(loc=1.0, scale=std_r * (1.0 + depth ** 2.0), size=N)
            local = np.random.normal(loc=1.0, scale=(std_l * (1.0 + depth ** 1.5))[None, :], size=(R, N))
            # and the whole thing through plot_beam_frame itself
            np.random.seed(1000 + p)
            rr = ref.UltrasoundRenderer(num_samples=S, attenuation_coeff=1e-3)
            with quiet():
                _, _, _, full = rr.plot_beam_frame(volume=v, source=torch.from_numpy(s[p]), directions=torch.from_numpy(d[p]),
                                                   plot=False, artifacts=True, std_radial=std_r, std_local=std_l,
                                                   max_sigma=max_sigma, alpha=alpha, start=start)
            import matplotlib.pyplot as plt
            plt.close("all")
            out[f"{tag}_pose"] = np.int64(p); out[f"{tag}_S"] = np.int64(S); out[f"{tag}_start"] = np.int64(start)
            out[f"{tag}_params"] = np.array([std_r, std_l, max_sigma, alpha])
            out[f"{tag}_frame"] = f.numpy(); out[f"{tag}_radial"] = radial; out[f"{tag}_local"] = local
            out[f"{tag}_speckle"] = np.asarray(a1); out[f"{tag}_blur"] = np.asarray(a2); out[f"{tag}_sharp"] = np.asarray(a3)
            out[f"{tag}_full"] = np.asarray(full); out[f"{tag}_full_dtype"] = np.array(str(full.dtype))
        out["tags"] = np.array([c[0] for c in cases])
        save("g14_artifacts", **out)

    # ---- G15: impedance MLP (src/impedance.py:6-17) and compute_impedance_volume (:38-53) ----------
    if want("g15"):
        import importlib
        imp = importlib.import_module("src.impedance")
        utl = importlib.import_module("src.utils")
        out = {}
        torch.manual_seed(77)
        m = imp.ImpedanceEstimator(1)
        with torch.no_grad():           # a trained net has biases / weights of both signs and sizes
            for q in m.parameters():
                q.mul_(1.7)
            m.model[4].bias.add_(1.5)
        for k, v_ in m.state_dict().items():
            out["sd_" + k] = v_.numpy().copy()
        g = torch.Generator().manual_seed(5)
        x = (torch.randn(3001, 1, generator=g) * 1.5)
        up = torch.randn(3001, 1, generator=g)
        with torch.enable_grad():
            xr = x.clone().requires_grad_(True)
            y = m(xr)
            (y * up).sum().backward()
        out["x"] = x.numpy(); out["up"] = up.numpy(); out["y"] = y.detach().numpy(); out["gx"] = xr.grad.numpy()
        for k, q in m.named_parameters():
            out["g_" + k] = q.grad.numpy().copy()
        # a synthetic "MRI": bright head on dark noisy air, with holes and specks the open/close cleans up
        n0, n1, n2 = 40, 36, 33
        u0 = (np.arange(n0) / (n0 - 1) - 0.5)[:, None, None]; u1 = (np.arange(n1) / (n1 - 1) - 0.5)[None, :, None]
        u2 = (np.arange(n2) / (n2 - 1) - 0.5)[None, None, :]
        rng = np.random.default_rng(15)
        mri = rng.uniform(0, 30, size=(n0, n1, n2))
        head = (u0 / 0.42) ** 2 + (u1 / 0.40) ** 2 + (u2 / 0.55) ** 2 <= 1.0     # touches the dim-2 border
        mri[head] = 300 + 200 * np.sin(9 * u0 + 5 * u1 + 7 * u2)[head] + rng.normal(0, 20, size=int(head.sum()))
        mri[rng.uniform(size=mri.shape) < 0.02] = 5.0         # holes inside, filled by the dilation
        mri[rng.uniform(size=mri.shape) < 0.01] = 500.0       # specks outside, removed by the erosion
        mri = mri.astype(np.float32)
        for thr in (50, 120.5):
            mask = utl.create_brain_mask(mri, thr)
            vn = utl.zscore_normalize(torch.from_numpy(mri), mask)
            Z = imp.ImpedanceEstimator.compute_impedance_volume(torch.from_numpy(mri), m, thr)
            tag = "t%d" % int(thr)
            out[tag + "_thr"] = np.float64(thr); out[tag + "_mask"] = mask.numpy(); out[tag + "_vnorm"] = vn.numpy()
            out[tag + "_Z"] = Z.numpy()
        out["mri"] = mri
        save("g15_impedance", **out)

    # ---- G16: the other functions `from src.renderer import *` hands to the notebooks --------------------------
    # prop_single_ray (full w vector), propagate_full_rays_batched (cumulated series), custom_nearest_sampler on
    # arbitrary points, and the simulate_rays / trace_ray methods themselves
    if want("g16"):
        g = torch.Generator().manual_seed(16)
        out = {}
        r = (torch.rand(7, 23, generator=g) - 0.5) * 1.2
        r[2, 5] = float("nan")                       # a NaN anywhere: the whole solution of that ray is zeroed
        r[3, 4] = 1.0; r[3, 9] = -1.0                # air interfaces
        r[4] = 0.0
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            rr = r.to(dt)
            out["w_" + tag] = ref.prop_single_ray(rr).numpy()
            out["cum_" + tag] = ref.propagate_full_rays_batched(rr).numpy()
        out["r"] = r.numpy()
        out["w_empty"] = ref.prop_single_ray(torch.zeros(3, 0)).numpy()
        vol = torch.rand(9, 11, 13, generator=g) * 1e6 + 1e6
        pts = torch.rand(5, 17, 3, generator=g) * torch.tensor([12.0, 14.0, 16.0]) - 2.0     # inside and outside
        pts[0, :6, 0] = torch.tensor([0.5, 1.5, 2.5, 3.5, -0.5, 8.5])                      # ties: round half to even
        with quiet():
            x, y, z, v = ref.custom_nearest_sampler(vol, pts, visualize=False)
            x64, y64, z64, v64 = ref.custom_nearest_sampler(vol, pts.double(), visualize=False)
        out.update(vol=vol.numpy(), pts=pts.numpy(), sx=x.numpy(), sy=y.numpy(), sz=z.numpy(), sv=v.numpy())
        assert torch.equal(x, x64) and torch.equal(v, v64)
        v32 = torch.from_numpy(phantom(32))
        s, d = pose_ring(32, 4, 12)
        rrn = ref.UltrasoundRenderer(num_samples=40, attenuation_coeff=1e-3)
        with quiet():
            tx, ty, tz, tv = ref.UltrasoundRenderer.trace_ray(v32, torch.from_numpy(s[1]), torch.from_numpy(d[1]), 40, 0)
            qx, qy, qz, qr = rrn.simulate_rays(v32, torch.from_numpy(s[1]), torch.from_numpy(d[1]), start=7)
            z1 = rrn.simulate_rays(v32, torch.from_numpy(s[1]), torch.from_numpy(d[1]), num_samples=25, MRI=True)
        import matplotlib.pyplot as plt
        plt.close("all")
        out.update(pose_src=s[1], pose_dirs=d[1], tr_x=tx.numpy(), tr_y=ty.numpy(), tr_z=tz.numpy(), tr_v=tv.numpy(),
                   sim_x=qx.numpy(), sim_y=qy.numpy(), sim_z=qz.numpy(), sim_r=qr.numpy(), sim_mri=z1.numpy())
        save("g16_api_functions", **out)

    # ---- G17: ILL-CONDITIONED rays of the benchmark workload (config 3), through the reference's dense solves ----------
    # Rays that graze the skull cross air <-> bone several times: |r| -> 0.9997, the transfer matrices are nearly
    # singular and ANY float32 evaluation (the reference's LU included) carries cond x eps of noise.  Recorded here so
    # that the tolerance of the full-size parity tests is the reference's own float32 noise on exactly these rays, not a
    # guess: impedance along the ray (trilinear samples of the 256^3 phantom, from the oracle -- the reference has no
    # trilinear sampler) -> the reference's compute_reflection_coeff + compute_echo_traces in fp32 and fp64.
    if want("g17"):
        from oracle import oracle as orc
        v = phantom(256)
        s, d = pose_ring(256, 32, 256)
        picks = [(18, 4), (30, 22), (0, 128)]            # two grazing rays, one ordinary ray
        Z = np.stack([orc.sample_trilinear(v, s[p], d[p][r:r + 1], 512)[0] for p, r in picks])
        Z32 = torch.from_numpy(Z)
        r32 = ref.UltrasoundRenderer.compute_reflection_coeff(Z32[:, :-1], Z32[:, 1:])
        with quiet():
            e32, _ = ref.compute_echo_traces(r32)
            e64, _ = ref.compute_echo_traces(r32.double())
        save("g17_grazing_rays", picks=np.array(picks, dtype=np.int64), Z=Z, r=r32.numpy(), echo32=e32.numpy(),
             echo64=e64.numpy())

    # ---- G18: the reference's OWN autograd through compute_echo_traces / propagate_full_rays_batched -------------
    # (N+1 LinalgSolveBackward nodes each; fp64 so that the values are the algorithm's, not the LU's noise)
    if want("g18"):
        g = torch.Generator().manual_seed(18)
        r = (torch.rand(5, 40, generator=g, dtype=torch.float64) - 0.5) * 1.2
        r[1, 17] = 0.9995                      # a nearly singular interface
        r[2, 25:] = 0.0                        # homogeneous tail
        r[3, 11] = float("nan")                # nan_to_num(nan=0) zeroes every echo from here on (:408)
        w_e = torch.linspace(0.3, 1.7, 41, dtype=torch.float64)[None, :] * torch.tensor([1.0, -0.5, 2.0, 1.0, 0.25], dtype=torch.float64)[:, None]
        out = {"r": r.numpy(), "w": w_e.numpy()}
        with torch.enable_grad():
            for name, fn in (("echo", lambda x: ref.compute_echo_traces(x)[0]), ("prop", ref.propagate_full_rays_batched)):
                x = r.clone().requires_grad_(True)
                with quiet():
                    y = fn(x)
                (y * w_e).sum().backward()
                out[name] = y.detach().numpy()
                out["g_" + name] = x.grad.numpy()
        save("g18_echo_autograd", **out)

    # ---- G20: rasterize_fan (src/renderer.py:626-653; host-side SciPy griddata) on a small scattered fan ----------
    # ---- G21: mri_to_us_point / us_to_mri_point of src/cone.py:21-59 (executed from the file via ast, like G12) -------
    if want("g21"):
        tree = ast.parse(open(os.path.join(REF, "src", "cone.py")).read())
        names = ("voxel_to_world", "world_to_voxel", "mri_to_us_point", "us_to_mri_point")
        fns = [n_ for n_ in tree.body if isinstance(n_, ast.FunctionDef) and n_.name in names]
        ns = {"np": np, "torch": torch}
        exec(compile(ast.Module(body=fns, type_ignores=[]), "cone.py", "exec"), ns)
        rng = np.random.default_rng(21)
        def aff(scale, shift):
            A = np.eye(4); q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            # a mild rotation about the identity, so that points stay inside both volumes
            R = np.eye(3) + 0.05 * (q - q.T); A[:3, :3] = R * scale; A[:3, 3] = shift
            return A
        T1 = rng.normal(size=(24, 20, 16)).astype(np.float32); US = rng.normal(size=(30, 26, 22)).astype(np.float32)
        At, Au = aff(1.0, (-3.0, 2.0, 1.0)), aff(0.8, (-1.0, 0.5, -0.5))
        out = {"T1": T1, "US": US, "At": At, "Au": Au}
        pts = [(5, 7, 4), (12, 3, 9), (20, 15, 12)]
        for j, (i, jj, k) in enumerate(pts):
            sl, idx = ns["mri_to_us_point"](i, jj, k, T1, At, US, Au)
            out[f"m2u{j}_in"] = np.array([i, jj, k]); out[f"m2u{j}_slice"] = sl; out[f"m2u{j}_idx"] = idx
        for j, (i, jj, k) in enumerate([(6, 8, 10), (14, 12, 15), (3, 20, 5)]):
            sl, idx = ns["us_to_mri_point"](i, jj, k, US, Au, T1, At)
            out[f"u2m{j}_in"] = np.array([i, jj, k]); out[f"u2m{j}_slice"] = sl; out[f"u2m{j}_idx"] = idx
        out["n"] = np.int64(3)
        save("g21_point_maps", **out)

    if want("g20"):
        g = np.random.default_rng(20)
        ang = g.uniform(-0.5, 0.5, 60)
        rad = g.uniform(5.0, 40.0, 60)
        xs, zs = rad * np.sin(ang), rad * np.cos(ang)
        val = g.uniform(0.0, 1.0, 60)
        img = ref.rasterize_fan(xs, zs, val)
        save("g20_rasterize_fan", x=xs, z=zs, v=val, img=img)

    # ---- G19 (--big): EVERY ray of the ill-conditioned poses of config 3 through the reference's dense solves -----
    # the widened tolerance of the full-size tests (oracle/conditioning.py) is a model; this pins
    # it on /root/reference/src/renderer.py:407 itself.  Per pose: impedance along all 256 rays (trilinear samples of
    # the 256^3 phantom from the oracle -- the reference has no trilinear sampler) -> the reference's
    # compute_reflection_coeff + compute_echo_traces in fp32 AND fp64 (~2.5 + ~5 minutes of dense solves per pose).
    if args.big and want("g19"):
        import time
        from oracle import oracle as orc
        v = phantom(256)
        s, d = pose_ring(256, 32, 256)
        path = os.path.join(HERE, "g19_ill_conditioned_poses.npz")
        out = dict(np.load(path)) if os.path.exists(path) else {}      # resumable: ~8 minutes of solves per (pose, precision)
        out["poses"] = np.array([18, 6], dtype=np.int64)
        for p in (18, 6):
            Z = orc.sample_trilinear(v, s[p], d[p], 512)
            Z32 = torch.from_numpy(Z)
            r32 = ref.UltrasoundRenderer.compute_reflection_coeff(Z32[:, :-1], Z32[:, 1:])
            out[f"Zsum_{p}"] = np.float64(Z.astype(np.float64).sum())
            out[f"r_{p}"] = r32.numpy()
            # echo32: the reference's float32 pipeline from the float32 samples (r in f32, dense solves in f32);
            # echo64: float64 solves from THAT float32 r (the solver's noise alone);
            # echo64z: the float64 pipeline from the same samples (r formed in f64 too: what every float32 evaluation,
            #          the reference's included, is an approximation of)
            for key, fn in ((f"echo32_{p}", lambda: ref.compute_echo_traces(r32)[0]),
                            (f"echo64_{p}", lambda: ref.compute_echo_traces(r32.double())[0]),
                            (f"echo64z_{p}", lambda: ref.compute_echo_traces(
                                ref.UltrasoundRenderer.compute_reflection_coeff(Z32[:, :-1].double(), Z32[:, 1:].double()))[0])):
                if key in out:
                    continue
                t0 = time.time()
                with quiet():
                    out[key] = fn().numpy()
                print(f"g19 {key}: {time.time() - t0:.0f} s", flush=True)
                np.savez_compressed(path, **out)
        save("g19_ill_conditioned_poses", **out)

    # ---- G10 (--big): config-2 shape forward, 256 rays x 512 steps ---------------
    if args.big and want("g10"):
        v = torch.from_numpy(phantom(256))
        s, d = pose_ring(256, 32, 256)
        x, y, z, f = ref_frame(v, torch.from_numpy(s[0]), torch.from_numpy(d[0]), 512, 1e-4, 0)
        save("g10_config2_fwd", pose=np.int64(0), P=np.int64(32), source=s[0], directions=d[0], frame=f.astype(np.float32))
    return 0


if __name__ == "__main__":
    sys.exit(main())
