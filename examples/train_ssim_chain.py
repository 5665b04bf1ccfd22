#!/usr/bin/env python3
"""`train_step` of the reference's `[DEMO] Train MRI to Impedance MLP - GPU` notebook (cell 16), the whole chain:

    MLP(MRI slice) -> impedance slice into the volume -> plot_beam_frame (64 rays x 228 samples, start 110)
      -> rotate_around_apex -> differentiable_splat(256 x 256, sigma = 0.5) -> min-max normalise -> 1 - SSIM
      -> backward -> Adam

on the HIP path (reference src/renderer.py:201-275, :655-692, :694-737).  The reference cannot run it with its current
source and, where it could, syncs with the host in every step (three `.item()` in differentiable_splat,
:704; `loss.item()`).  Here nothing in the iteration touches the host -- the splat picks its axes on the device
(diffus_splat_axes) -- so ONE captured hipGraph replays it.

    python examples/train_ssim_chain.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import diffus_amd as da  # noqa: E402
from diffus_amd.phantom import phantom, pose_ring  # noqa: E402
from losses import gaussian_window, minmax01, ssim  # noqa: E402


class SsimLoop:
    def __init__(self, n=256, rays=64, samples=228, start=110, alpha=1e-4, lr=1e-2, pose=0, seed=0, image=256, sigma=0.5,
                 fused_loss=True):
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
        self.image, self.sigma = image, sigma
        # fused_loss: min-max + 1 - SSIM as one HIP autograd node (diffus_amd.ssim_loss); else the same thing as torch ops
        self.fused_loss = fused_loss
        # the sample coordinates depend on the pose only: the index planes of one plot_beam_frame call (nearest sampling
        # = the reference's sampler), as float tensors like the notebook's `x.float()`
        rend = da.UltrasoundRenderer(samples, alpha)
        x, y, z, _ = rend.plot_beam_frame(z_true, s[0], d[0], start=start)
        self.x, self.y, self.z = x.flatten().float(), y.flatten().float(), z.flatten()
        self.apex = torch.tensor([image / 2.0, 8.0], device=dev)         # where the fan's apex goes in the image
        mid = d[0, rays // 2, :2]
        self.median = (mid / mid.norm()).to(dev)                         # device tensors: nothing is copied per step
        self.window = gaussian_window(device=dev)
        self.step = da.CapturedStep(z_true.clone(), s, d, samples, alpha, "nearest", start=start, layout="canonical",
                                    alias_grads=True)
        self.real = minmax01(self.splat_of(self.frame_of_truth())).detach()   # the "real" ultrasound image, normalised
        self.loss = torch.zeros((), device=dev)
        self.one = torch.ones((), device=dev)
        self.graph, self.repeat = None, 1

    def frame_of_truth(self):
        self.step.fwd()
        return self.step.frame.clone()

    def splat_of(self, frame):
        """frame (1,R,N1) -> (W,H) image: rotate_around_apex -> differentiable_splat."""
        xr, yr = da.rotate_around_apex(self.x, self.y, self.apex, self.median)
        return da.differentiable_splat(xr, yr, self.z, frame.reshape(-1), H=self.image, W=self.image, sigma=self.sigma)

    def loss_of(self, z_slice):
        frame = self.step.render(self.step.volume_with_slice(z_slice, 2, self.k))
        img = self.splat_of(frame)
        if self.fused_loss:
            return da.ssim_loss(img, self.real)                          # min-max normalisation + 1 - SSIM, one node
        return 1.0 - ssim(minmax01(img)[None, None], self.real[None, None], data_range=1.0, window=self.window)

    def iteration(self):
        # the prediction lands in its slice of the step's volume (no copy launch), its gradient is read from there
        loss = self.loss_of(self.model(self.mri, scale=1e6, out=self.step.slice_view(2, self.k)))
        self.opt.zero_grad(set_to_none=True)
        loss.backward(self.one)             # a resident 1.0: `backward()` alone fills a fresh one every iteration (a launch)
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
    loop = SsimLoop()
    loop.iteration()
    first = float(loop.loss)
    ms_eager = loop.run(20)
    loop.capture()
    ms = loop.run(steps)
    loop.capture(repeat=8)
    ms8 = loop.run(steps)
    print(f"1 - SSIM {first:.4f} -> {float(loop.loss):.4f} after {2 * steps + 30} iterations; {ms_eager:.3f} ms per iteration eager, "
          f"{ms:.3f} ms as one captured hipGraph per iteration, {ms8:.3f} ms with 8 iterations per graph")
