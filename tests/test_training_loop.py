"""The reference's MLP training loop (`[DEMO] Train MRI to Impedance MLP - GPU` cell 16: MLP -> impedance slice ->
plot_beam_frame -> loss -> backward -> Adam) through CapturedStep.render: gradients equal the drop-in autograd path's,
the loss goes down, and one iteration -- captured as a single hipGraph -- stays under 0.15 ms."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

pytestmark = pytest.mark.gpu


def test_render_node_matches_the_drop_in_autograd_path():
    import diffus_amd as da
    from diffus_amd.phantom import phantom, pose_ring
    n, P, R, S, start, alpha = 64, 3, 24, 120, 30, 2e-3
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 8, R)
    for sampler, layout in (("trilinear", "paired"), ("nearest", "canonical")):
        v = vol.clone().requires_grad_(True)
        s = torch.from_numpy(src[:P]).cuda().requires_grad_(True)
        d = torch.from_numpy(dirs[:P]).cuda().requires_grad_(True)
        step = da.CapturedStep(v, s, d, S, alpha, sampler, start=start, layout=layout)
        w = torch.linspace(0.5, 2.0, S - start, device="cuda")
        f = step.render(v, s, d)
        ((f * w) ** 2).sum().backward()
        got = (f.detach().clone(), v.grad.clone(), s.grad.clone() if s.grad is not None else None,
               d.grad.clone() if d.grad is not None else None)
        v2 = vol.clone().requires_grad_(True)
        s2 = torch.from_numpy(src[:P]).cuda().requires_grad_(True)
        d2 = torch.from_numpy(dirs[:P]).cuda().requires_grad_(True)
        f2 = da.render_poses(v2, s2, d2, S, alpha, start=start, sampler=sampler, layout=layout)
        ((f2 * w) ** 2).sum().backward()
        assert torch.equal(got[0], f2.detach())
        assert float((got[1] - v2.grad).abs().max()) <= 2e-5 * float(v2.grad.abs().max())
        if sampler == "trilinear":
            assert float((got[2] - s2.grad).abs().max()) <= 1e-5 * float(s2.grad.abs().max())
            assert float((got[3] - d2.grad).abs().max()) <= 1e-5 * float(d2.grad.abs().max())
        # a stale frame cannot be back-propagated once a later render() has reused the buffers
        fa = step.render(v, s, d)
        step.render(v, s, d)
        with pytest.raises(RuntimeError):
            (fa ** 2).sum().backward()


def test_mse_loss_node_matches_the_drop_in_autograd_path():
    """CapturedStep.mse_loss: the one-pass step as an autograd node (forward = frame + loss + gradients, backward = a
    scaling) against render_poses + torch's mse_loss, for a whole volume with poses and for a slice of it."""
    import diffus_amd as da
    from diffus_amd.phantom import phantom, pose_ring
    n, P, R, S, start, alpha = 64, 3, 24, 120, 30, 2e-3
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 8, R)
    g = torch.Generator().manual_seed(5)
    target = (torch.randn(P, R, S - start, generator=g) * 0.02).cuda()
    for sampler, layout in (("trilinear", "paired"), ("nearest", "canonical")):
        v = vol.clone().requires_grad_(True)
        s = torch.from_numpy(src[:P]).cuda().requires_grad_(True)
        d = torch.from_numpy(dirs[:P]).cuda().requires_grad_(True)
        step = da.CapturedStep(v, s, d, S, alpha, sampler, start=start, layout=layout)
        step.set_target(target, 1.0 / target.numel())
        loss = step.mse_loss(v, s, d)
        (3.0 * loss).backward()                         # an upstream factor reaches all three gradients
        v2 = vol.clone().requires_grad_(True)
        s2 = torch.from_numpy(src[:P]).cuda().requires_grad_(True)
        d2 = torch.from_numpy(dirs[:P]).cuda().requires_grad_(True)
        f2 = da.render_poses(v2, s2, d2, S, alpha, start=start, sampler=sampler, layout=layout)
        loss2 = torch.nn.functional.mse_loss(f2, target)
        (3.0 * loss2).backward()
        assert float((step.frame - f2.detach()).abs().max()) <= 2e-5 * float(f2.detach().abs().max())
        assert abs(float(loss) - float(loss2)) <= 1e-5 * abs(float(loss2))
        assert float((v.grad - v2.grad).abs().max()) <= 1e-4 * float(v2.grad.abs().max())
        if sampler == "trilinear":
            assert float((s.grad - s2.grad).abs().max()) <= 1e-4 * float(s2.grad.abs().max())
            assert float((d.grad - d2.grad).abs().max()) <= 1e-4 * float(d2.grad.abs().max())
    # a slice of the volume as the learnable (the reference's training notebook, cell 16)
    k = int(round(float(src[0, 2])))                    # the plane the first fan lies in
    step = da.CapturedStep(vol.clone(), torch.from_numpy(src[:P]).cuda(), torch.from_numpy(dirs[:P]).cuda(), S, alpha,
                           "nearest", start=start, layout="canonical", persistent=False)
    step.set_target(target, 1.0 / target.numel())
    sl = (vol[:, :, k] * 1.02).clone().requires_grad_(True)
    step.mse_loss(slice_values=sl, slice_dim=2, slice_index=k).backward()
    v2 = vol.clone()
    sl2 = (vol[:, :, k] * 1.02).clone().requires_grad_(True)
    v2[:, :, k] = sl2
    f2 = da.render_poses(v2, torch.from_numpy(src[:P]).cuda(), torch.from_numpy(dirs[:P]).cuda(), S, alpha, start=start,
                         sampler="nearest", layout="canonical")
    torch.nn.functional.mse_loss(f2, target).backward()
    assert float(sl2.grad.abs().max()) > 0
    assert float((sl.grad - sl2.grad).abs().max()) <= 1e-4 * float(sl2.grad.abs().max())
    with pytest.raises(RuntimeError):                   # a stale loss cannot be back-propagated
        la = step.mse_loss(slice_values=sl, slice_dim=2, slice_index=k)
        step.mse_loss(slice_values=sl, slice_dim=2, slice_index=k)
        la.backward()


@pytest.mark.parametrize("one_pass", [True, False])
def test_mlp_training_loop_as_one_captured_graph(one_pass):
    from train_impedance_mlp import Loop
    loop = Loop(one_pass=one_pass)
    loop.iteration()
    torch.cuda.synchronize()
    first = float(loop.loss)
    for _ in range(30):
        loop.iteration()
    eager = float(loop.loss)
    assert eager < first                              # it learns
    loop.capture()
    ms = loop.run(200)
    last = float(loop.loss)
    assert last < eager and last == last
    print(f"captured MLP training iteration (one_pass={one_pass}): {ms:.3f} ms, loss {first:.3e} -> {last:.3e}")
    assert ms <= 0.15, ms
