#!/usr/bin/env python3
"""The training loop of the reference's `[DEMO] Train MRI to Impedance MLP - GPU` notebook (cells 16-18) on the HIP
path: an MLP maps the MRI intensities of the imaging plane to acoustic impedance, the frame is rendered through that
impedance (`plot_beam_frame`, 64 rays x 228 samples, start = 110), a loss compares it with the target frame, Adam updates
the MLP.  The reference cannot run this loop with its current source; here the whole iteration -- fused MLP
forward, slice update, median + forward frame, loss, adjoint scan + volume scatter, MLP backward, fused Adam -- is ONE
captured hipGraph replayed per step.  By default the MSE is fused into the renderer (`CapturedStep.mse_loss`: frame, loss and
d/dimpedance out of ONE pass over the samples); `Loop(one_pass=False)` renders the frame as an autograd node and
leaves the loss to torch.

    python examples/train_impedance_mlp.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffus_amd as da  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402


class Loop:
    def __init__(self, n=256, rays=64, samples=228, start=110, alpha=1e-4, lr=1e-2, pose=0, seed=0, one_pass=True):
        dev = torch.device("cuda", torch.cuda.current_device())
        torch.manual_seed(seed)
        z_true = torch.from_numpy(phantom(n)).to(dev)                    # "ground truth" impedance
        src, dirs = pose_ring(n, 32, rays)
        self.k = int(round(float(src[pose, 2])))                         # the imaging plane (fans lie in a dim-2 plane)
        s = torch.from_numpy(src[pose:pose + 1]).to(dev)
        d = torch.from_numpy(dirs[pose:pose + 1]).to(dev)
        s[0, 2] = float(self.k)
        self.mri = (z_true[:, :, self.k] / 1e6).contiguous()             # stand-in MRI intensities of that plane
        self.model = da.ImpedanceEstimator().to(dev)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, capturable=True, fused=True)
        # nearest sampling = the reference's sampler; canonical layout: only one slice of the volume changes per step (the
        # kernels read the tensor in place; the gradient still comes back through the sparse, never-memset hand-back)
        self.step = da.CapturedStep(z_true.clone(), s, d, samples, alpha, "nearest", start=start, layout="canonical",
                                    alias_grads=True)
        self.step.fwd()
        self.target = self.step.frame.clone()                            # the frame of the true impedance
        # one_pass: MSE (mean over the frame, like torch's mse_loss) fused into the renderer -- frame, loss and gradient
        # come out of one call (CapturedStep.mse_loss); else the frame is a node and the loss is torch's
        self.one_pass = one_pass
        self.step.set_target(self.target, 1.0 / self.target.numel())
        self.loss = torch.zeros((), device=dev)
        self.one = torch.ones((), device=dev)
        self.graph, self.repeat = None, 1

    def iteration(self):
        # the prediction lands in its slice of the step's volume (no copy launch), its gradient is read from there
        z_slice = self.model(self.mri, scale=1e6, out=self.step.slice_view(2, self.k))
        if self.one_pass:
            loss = self.step.mse_loss(slice_values=z_slice, slice_dim=2, slice_index=self.k)
        else:
            frame = self.step.render(self.step.volume_with_slice(z_slice, 2, self.k))
            loss = torch.nn.functional.mse_loss(frame, self.target)
        self.opt.zero_grad(set_to_none=True)
        # a resident 1.0 as the upstream gradient: `backward()` alone fills a fresh one every iteration (a launch), and the
        # step's own unit lets mse_loss hand its gradients over unscaled (no multiply launches)
        loss.backward(self.step.unit if self.one_pass else self.one)
        self.opt.step()
        self.loss = loss.detach()           # no copy: inside a captured graph this tensor is rewritten by every replay

    def capture(self, repeat=1):
        """One hipGraph of `repeat` whole iterations (two graph LAUNCHES are ~8.6 us apart on this stack, kernels inside one
        graph are not: several iterations per graph amortise that)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self.iteration()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph, self.repeat = torch.cuda.CUDAGraph(), int(repeat)
        with torch.cuda.graph(self.graph):
            for _ in range(self.repeat):
                self.iteration()
        return self.graph

    def run(self, steps):
        """`steps` iterations (rounded up to whole graphs) -> ms per iteration."""
        per = self.repeat if self.graph is not None else 1
        calls = -(-steps // per)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            if self.graph is not None:
                self.graph.replay()
            else:
                self.iteration()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / (calls * per)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    loop = Loop()
    loop.iteration()
    first = float(loop.loss)
    ms_eager = loop.run(20)
    loop.capture()
    ms = loop.run(steps)
    loop.capture(repeat=8)
    ms8 = loop.run(steps)
    print(f"loss {first:.4e} -> {float(loop.loss):.4e} after {2 * steps + 30} iterations; {ms_eager:.3f} ms per iteration eager, "
          f"{ms:.3f} ms as one captured hipGraph per iteration, {ms8:.3f} ms with 8 iterations per graph")
