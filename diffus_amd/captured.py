"""CapturedStep -- the hot path as a training loop runs it: persistent buffers, direct C-ABI calls, one hipGraph.

`render_poses` + torch autograd (renderer.py) is the drop-in path: it allocates its outputs, goes through the
autograd engine and costs ~0.1-0.2 ms of HOST time per step whatever the batch.  A loop that renders the same
number of poses every iteration (the reference's `[DEMO] Train MRI to Impedance MLP - GPU` cell 16, or a pose
registration) does not need any of that: every buffer can be allocated once and the whole step -- forward
(reference src/renderer.py:201-275), loss, backward, gradient hand-back -- replayed as one
captured hipGraph, because libdiffus_hip.so never allocates or synchronises.

    step = CapturedStep(volume, sources, directions, num_samples=512, attenuation_coeff=1e-4)
    step.capture()
    for it in range(iters):
        step.set_poses(new_sources, new_directions)      # in place: the graph reads the same buffers
        step.replay()                                    # forward + sum-of-squares loss + backward + hand-back
        use(step.loss, step.gvol, step.gsrc, step.gdirs) # (P,), (d0,d1,d2), (P,3), (P,R,3)

With an external loss (splat -> SSIM, ...) either run `forward()`, write dL/dframe into `step.gframe`, then
`backward()` (both halves capture separately), or call `step.render(volume, sources, directions)`: it returns the
frame as a node of torch's autograd graph, so `loss.backward()` reaches the volume and the poses through the captured
backward -- the drop-in `render_poses` costs 0.1-0.3 ms of host time per step, this ~0.05 ms.  A learnable volume (`learnable_volume=True`) is re-converted to
the kernels' layout inside every step, so an optimiser may update `volume` in place between replays.

The volume gradient comes back in the caller's canonical (d0,d1,d2) tensor `gvol`.  By default it is PERSISTENT:
never memset, `diffus_gradbuf_flush(PERSISTENT)` stores the bricks this step touched and clears the ones only the
previous step touched, so after every step `gvol` is exactly this step's dense gradient.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .renderer import _stream_id

_EMPTY_BOX = ((0, 0), (0, 0), (0, 0))
_LAYOUT_ID = {"canonical": _lib.CANONICAL, "bricked": _lib.BRICKED, "paired": _lib.PAIRED}
_SAMPLER_ID = {"nearest": _lib.NEAREST, "prop": _lib.NEAREST, "trilinear": _lib.TRILINEAR}


def _vp(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None        # ctypes turns the int into the void* of the signature


def _pose_tensor(t: torch.Tensor, dev: torch.device) -> torch.Tensor:
    """f64 stays f64 (it selects the reference's rounding sequence, src/renderer.py:119-124), anything else -> f32;
    a tensor that is already resident, contiguous and of that dtype is used as is, so the caller can update it in place."""
    dt = torch.float64 if t.dtype == torch.float64 else torch.float32
    t = t.detach()
    if t.device != dev or t.dtype != dt:
        t = t.to(device=dev, dtype=dt)
    return t if t.is_contiguous() else t.contiguous()


class _CapturedRender(torch.autograd.Function):
    """frame = step.forward(); backward = step.backward() with the incoming dL/dframe.  The step owns every buffer."""

    @staticmethod
    def forward(ctx, step, volume, sources, directions):
        step._run("forward")
        ctx.step = step
        ctx.stamp = step._stamp = step._stamp + 1
        ctx.shapes = (sources.shape, directions.shape)   # (3,) / (R,3) for one pose: gradients go back in the callers' shapes
        return step.frame.detach()          # a new tensor object on the step's frame buffer (no copy)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gframe):
        step = ctx.step
        if ctx.stamp != step._stamp:
            raise RuntimeError("CapturedStep.render: backward of a frame whose buffers a later render() has overwritten "
                               "(one forward, one backward, in that order)")
        own = step.gframe
        if gframe.data_ptr() != own.data_ptr():
            # eager launches take dL/dframe where it lies (no copy); a captured backward graph holds the step's own buffer
            if (step._graphs.get("backward") is None and gframe.is_contiguous() and gframe.dtype == own.dtype
                    and gframe.device == own.device and gframe.numel() == own.numel()):
                step.gframe = gframe
            else:
                own.copy_(gframe)
        try:
            step._run("backward")
        finally:
            step.gframe = own
        need_v, need_s, need_d = ctx.needs_input_grad[1:4]
        keep = (lambda t: t) if step.alias_grads else (lambda t: t.clone())
        return (None, keep(step.gvol) if (need_v and step.gvol is not None) else None,
                keep(step.gsrc).reshape(ctx.shapes[0]) if need_s else None, keep(step.gdirs).reshape(ctx.shapes[1]) if need_d else None)


class _CapturedMSE(torch.autograd.Function):
    """loss = sum_p loss_scale * sum((frame_p - target_p)^2): the forward IS the whole one-pass step (frame, per-pose
    losses and the three gradients out of diffus_render_step_mse); the backward only scales what is already there."""

    @staticmethod
    def forward(ctx, step, volume, sources, directions, slice_values, dim, index):
        if slice_values is not None:
            sl = step.vol.select(dim, index)
            if not (slice_values.data_ptr() == sl.data_ptr() and slice_values.shape == sl.shape
                    and slice_values.stride() == sl.stride() and slice_values.dtype == sl.dtype):
                sl.copy_(slice_values)  # (values written straight into slice_view(dim, index) are already in place)
            step._mark_slice(dim, index)
        step._run("step")
        ctx.step = step
        ctx.where = None if slice_values is None else (dim, index)
        ctx.stamp = step._stamp = step._stamp + 1
        ctx.shapes = (sources.shape, directions.shape)   # as handed in (autograd would otherwise sum (1,R,3) down to (R,3): a launch)
        # (one pose: the loss IS the buffer's only element -- a view, valid until the next step, instead of a reduction)
        return step.loss.sum() if step.P > 1 else step.loss.detach().view(())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        step = ctx.step
        if ctx.stamp != step._stamp:
            raise RuntimeError("CapturedStep.mse_loss: backward of a loss whose buffers a later call has overwritten")
        need_v, need_s, need_d, need_sl = ctx.needs_input_grad[1:5]
        have_v = step.gvol is not None
        if g.data_ptr() == step.unit.data_ptr():
            # `loss.backward(step.unit)`: the upstream gradient IS the step's resident 1.0 -- the gradients are handed over as
            # they are (views of the step's buffers; the slice a strided one) instead of through four multiply launches
            return (None, step.gvol if (need_v and have_v) else None, step.gsrc.reshape(ctx.shapes[0]) if need_s else None,
                    step.gdirs.reshape(ctx.shapes[1]) if need_d else None,
                    step.gvol.select(*ctx.where) if (need_sl and have_v and ctx.where is not None) else None, None, None)
        return (None, step.gvol * g if (need_v and have_v) else None,
                (step.gsrc * g).reshape(ctx.shapes[0]) if need_s else None, (step.gdirs * g).reshape(ctx.shapes[1]) if need_d else None,
                step.gvol.select(*ctx.where) * g if (need_sl and have_v and ctx.where is not None) else None, None, None)


class _SliceIntoVolume(torch.autograd.Function):
    """step.vol[..., index, ...] = values (in place, the other voxels keep their values); d/dvalues = that slice of d/dvolume."""

    @staticmethod
    def forward(ctx, step, values, dim, index):
        sl = step.vol.select(dim, index)
        if not (values.data_ptr() == sl.data_ptr() and values.shape == sl.shape and values.stride() == sl.stride()
                and values.dtype == sl.dtype):
            sl.copy_(values)            # (values written straight into slice_view(dim, index) are already in place)
        step._mark_slice(dim, index)
        ctx.where = (dim, index)
        return step.vol.detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gvol):
        dim, index = ctx.where
        return None, gvol.select(dim, index), None, None      # a strided view: consumers that need it packed pack it


class CapturedStep:
    """Pre-allocated buffers + direct C-ABI calls; `capture()` turns `step()` into one hipGraph."""

    def __init__(self, volume: torch.Tensor, sources: torch.Tensor, directions: torch.Tensor, num_samples: int,
                 attenuation_coeff: float, sampler: str = "trilinear", start: int = 0, want_gvol: bool = True,
                 layout: str = "paired", sparse: bool = True, persistent: bool = True,
                 learnable_volume=False, alias_grads: bool = False, fused_loss: bool = True,
                 target: Optional[torch.Tensor] = None, loss_scale: float = 1.0, one_pass: bool = True,
                 bricked_grad: Optional[bool] = None, fans: str = "auto", repair_frames: bool = False):
        if not volume.is_cuda:
            raise _lib.DiffusError("CapturedStep needs a HIP-resident volume; there is no CPU fallback")
        if volume.dim() != 3 or volume.dtype != torch.float32 or not volume.is_contiguous():
            raise ValueError("volume must be a contiguous (d0,d1,d2) float32 tensor")
        self.lib = _lib.load()
        dev = volume.device
        self.dev = dev
        self.vol = volume
        self.dims = tuple(int(x) for x in volume.shape)
        self.src = _pose_tensor(sources, dev).reshape(-1, 3)
        self.dirs = _pose_tensor(directions, dev)
        if self.dirs.dim() != 3 or self.dirs.shape[0] != self.src.shape[0] or self.dirs.shape[2] != 3:
            raise ValueError(f"directions must be (P,R,3) with P = {self.src.shape[0]}; got {tuple(self.dirs.shape)}")
        self.layout = _LAYOUT_ID[layout]
        self.sampler = _SAMPLER_ID[sampler]
        # fans: "planar" -- the caller vouches that no ray moves along dim 2 (every fan of the reference: src/cone.py:258): the
        # volume scatter runs its launch for planar fans alone (DIFFUS_FANS_PLANAR; a wrong promise costs time, never
        # correctness); "oblique" -- the launch that also carries the slab path for fans that leave the slice; "auto" -- look
        # at the directions now (one readback) and again never: set_poses() with new directions makes the answer unknown,
        # and unknown means the slab-capable launch.
        # repair_frames: the one-pass step evaluates the frame rows of ILL-CONDITIONED rays (a ray grazing the skull: |echo| > 1)
        # again in float64 (DIFFUS_BWD_REPAIR_FRAME) -- what the forward kernel does by itself; here it costs ~10 us per step
        self.repair_frames = bool(repair_frames)
        if fans not in ("auto", "planar", "oblique"):
            raise ValueError("fans: 'auto', 'planar' or 'oblique'")
        self.fans = fans
        self._planar = (fans == "planar") or (fans == "auto" and bool((self.dirs[..., 2] == 0).all()))
        self.P, self.R = self.dirs.shape[0], self.dirs.shape[1]
        self.S, self.start, self.alpha = int(num_samples), int(start), float(attenuation_coeff)
        self.N1 = self.S - self.start
        self.learnable_volume = bool(learnable_volume)
        # learnable_volume: step() re-converts the volume first; all of it, or -- when the caller sets this to
        # ((x0, x1), (y0, y1), (z0, z1)) -- only the part it rewrites between steps (refresh_volume).
        # learnable_volume="slice": the volume changes ONLY through volume_with_slice() / mse_loss(slice_values=...) (the
        # reference's training loop: one slice of predicted impedance per iteration); those calls then set the box themselves.
        self._dirty_box = None
        self.slice_only = isinstance(learnable_volume, str) and learnable_volume == "slice"
        self.eager_fallbacks = 0     # replay()s that ran eagerly because the captured launches did not fit the inputs
        if isinstance(learnable_volume, str) and not self.slice_only:
            raise ValueError("learnable_volume: False, True or 'slice'")
        self._full_once = False      # a volume copied in by _adopt(): the next conversion covers all of it
        self._graph_box: dict = {}   # the box a captured "step" / "forward" graph has baked into its conversion launch
        # render(): hand autograd the step's own gradient buffers instead of clones of them.  Fine for the usual loop
        # (optimizer.zero_grad(set_to_none=True), one backward per forward); a .grad that autograd adopted is then
        # overwritten in place by the next step.
        self.alias_grads = bool(alias_grads)
        self._src_shape = tuple(sources.shape)
        self._stamp = 0
        # start > 0: the forward leaves the per-pose median (and the median ray's samples) in the workspace and the
        # backward may reuse it (DIFFUS_BWD_KEEP_MEDIAN) -- but only while it belongs to the CURRENT inputs.  `_median_of`
        # remembers the inputs' in-place version counters at the time the median was computed.
        self._median_of = None
        self._conv_stamp = 0
        # step(): loss_p = loss_scale * sum((frame_p - target_p)^2) (target None: the frame's energy).  fused_loss: the
        # backward forms dL/dframe from the frame on the fly and its closing per-pose blocks sum the loss
        # (diffus_render_bwd_mse) -- no loss kernel, no gradient-of-frame buffer traffic
        self.fused_loss = bool(fused_loss)
        # ... and one_pass: the frame too comes out of that backward (diffus_render_step_mse): the adjoint scan
        # recomputes the forward anyway, so step() needs no forward launch at all
        self.one_pass = bool(one_pass)
        self.target = target
        self.loss_scale = float(loss_scale)
        self.frame = torch.empty((self.P, self.R, self.N1), dtype=torch.float32, device=dev)
        self.gframe = torch.empty_like(self.frame)
        self.unit = torch.ones((), dtype=torch.float32, device=dev)     # `loss.backward(step.unit)`: see _CapturedMSE.backward
        d0, d1, d2 = self.dims
        # canonical gradient, what the caller gets.  persistent: the tensor is kept across steps and
        # diffus_gradbuf_flush(PERSISTENT) clears what the previous step left where this step adds nothing, so it
        # always equals this step's dense gradient without a 64 MiB memset per step.
        # The GRADIENT's layout is the bricked scratch + touched-brick flags for a bricked or paired volume, and by
        # default (bricked_grad=None) for a canonical one too (DIFFUS_GRAD_BRICKED): the kernels then read the caller's
        # tensor in place -- nothing to convert when the caller changes a slice of it every step -- and the gradient still
        # comes back sparsely, without a 64 MiB memset per step.  bricked_grad=False: scatter straight into the
        # canonical tensor (zeroed every step).
        self.grad_bricked = (self.layout != _lib.CANONICAL) or (want_gvol and sparse and bricked_grad is not False)
        self.persistent = persistent and sparse and want_gvol and self.grad_bricked
        self.gvol = torch.zeros_like(volume) if want_gvol else None
        nb = self.lib.diffus_bricked_floats(d0, d1, d2)
        if self.layout != _lib.CANONICAL:
            # HBM-resident converted copy of the volume
            nk = nb if self.layout == _lib.BRICKED else self.lib.diffus_paired_floats(d0, d1, d2)
            self.vol_k = torch.empty(nk, dtype=torch.float32, device=dev)
            self.refresh_volume()
            if self.slice_only:
                self.dirty_box = _EMPTY_BOX                    # nothing changes until a slice is written
        else:
            self.vol_k = volume
        if self.grad_bricked:
            # sparse gradient hand-back: the bricked scratch and its touched-brick flags are all-zero
            # between steps (diffus_gradbuf_flush restores that), only touched bricks are converted
            self.gvol_k = torch.zeros(nb, dtype=torch.float32, device=dev) if want_gvol else None
            self.touched = (torch.zeros(self.lib.diffus_brick_count(d0, d1, d2), dtype=torch.int32, device=dev)
                            if (want_gvol and sparse) else None)
        else:
            self.gvol_k, self.touched = self.gvol, None
        # nearest sampling has no pose gradient (integer indices, reference :754-758): the two tensors are zero once and
        # for all and are not handed to the library, which would otherwise memset them in every step
        self._pose_grads = self.sampler == _lib.TRILINEAR
        self.gsrc = (torch.empty if self._pose_grads else torch.zeros)((self.P, 3), dtype=torch.float32, device=dev)
        self.gdirs = (torch.empty if self._pose_grads else torch.zeros)((self.P, self.R, 3), dtype=torch.float32, device=dev)
        self.loss = torch.empty((self.P,), dtype=torch.float32, device=dev)
        self.loss_ws = torch.zeros(max(512 * self.P, 512), dtype=torch.uint8, device=dev)   # arrival counters: zero once
        nws = max(self.lib.diffus_workspace_bytes(self.P, self.R, self.S, self.start), 256)
        self.ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        sdt = _lib.DIFFUS_F64 if self.src.dtype == torch.float64 else _lib.DIFFUS_F32
        ddt = _lib.DIFFUS_F64 if self.dirs.dtype == torch.float64 else _lib.DIFFUS_F32
        # every pointer below is fixed for the life of the object (inputs are updated in place)
        self.common = (_vp(self.vol_k), d0, d1, d2, self.layout, _vp(self.src), sdt, _vp(self.dirs), ddt, self.P, self.R,
                       self.S, self.start, self.alpha, self.sampler)
        self._glay = self.layout | (_lib.GRAD_BRICKED if (self.grad_bricked and self.layout == _lib.CANONICAL) else 0)
        self._set_common_bwd()
        self._graphs: dict = {}
        self._side: Optional[torch.cuda.Stream] = None

    # -- plumbing ---------------------------------------------------------------------------------------------
    def stream(self):
        return _stream_id(self.dev)

    def _set_common_bwd(self):
        """Arguments of the backward entry points: the layout word carries the gradient's layout and the planar-fan hint."""
        self.common_bwd = self.common[:4] + (self._glay | (_lib.FANS_PLANAR if self._planar else 0),) + self.common[5:]

    @property
    def fans_planar(self) -> bool:
        """Whether the step launches the scatter for planar fans alone (see `fans`)."""
        return self._planar


    # -- inputs, updated in place (a captured graph keeps reading the same buffers) -----------------------------
    def _inputs_now(self):
        return (self.vol._version, self.src._version, self.dirs._version, self._conv_stamp)

    def _median_valid(self) -> bool:
        """The workspace median was computed from the inputs as they are now (no set_poses / in-place volume edit /
        refresh_volume since)."""
        return self.start > 0 and self._median_of is not None and self._median_of == self._inputs_now()

    def set_poses(self, sources: torch.Tensor, directions: Optional[torch.Tensor] = None):
        self.src.copy_(sources.reshape(self.src.shape))
        if directions is not None:
            self.dirs.copy_(directions.reshape(self.dirs.shape))
            self._directions_changed()
        self._stamp += 1            # a frame rendered before this call can no longer be back-propagated

    def _directions_changed(self):
        if self.fans == "auto" and self._planar:
            self._planar = False    # unknown from here on (no readback per step): the slab-capable launch; graphs captured
            self._set_common_bwd()  # before keep the planar launch, which stays correct for any fan (the general 3-D tile)

    @property
    def dirty_box(self):
        """The part of the canonical volume that changed since the last conversion, ((x0, x1), (y0, y1), (z0, z1)) half-open,
        or None (= all of it).  Always held as a tuple of int pairs, so that a box given as lists compares equal to the one a
        captured graph has baked in; checked against the volume's shape."""
        return self._dirty_box

    @dirty_box.setter
    def dirty_box(self, box):
        if box is None:
            self._dirty_box = None
            return
        box = tuple((int(a), int(b)) for a, b in box)
        if len(box) != 3 or any(not (0 <= a <= b <= n) for (a, b), n in zip(box, self.dims)):
            raise ValueError(f"dirty_box: three half-open ranges inside {self.dims}; got {box}")
        self._dirty_box = box

    def refresh_volume(self, box=None):
        """Rebuild the converted copy from `self.vol` (after the caller changed the volume in place).
        box = ((x0, x1), (y0, y1), (z0, z1)), half-open, or `self.dirty_box` when set: only that part of the canonical
        volume changed since the last conversion -- the reference's training loop rewrites one slice per step -- and only
        the records / bricks that hold it are rebuilt (diffus_convert_volume_box)."""
        self._conv_stamp += 1
        self._stamp += 1
        explicit = box is not None
        box = box if explicit else self.dirty_box
        if self._full_once:
            box, self._full_once = None, False
        if self.slice_only and not explicit:
            self._last_box = box                # what capture() bakes in when nothing is pending (the loop's slice)
            self._dirty_box = _EMPTY_BOX        # everything marked so far is converted by this call: nothing pending
        if box is not None and self.layout != _lib.CANONICAL:
            (x0, x1), (y0, y1), (z0, z1) = box
            _lib.check(self.lib.diffus_convert_volume_box(_vp(self.vol), *self.dims, self.layout, _vp(self.vol_k), int(x0), int(x1),
                                                          int(y0), int(y1), int(z0), int(z1), self.stream()),
                       "diffus_convert_volume_box")
            return
        if self.layout == _lib.BRICKED:
            _lib.check(self.lib.diffus_brick_volume(_vp(self.vol), *self.dims, _vp(self.vol_k), self.stream()), "diffus_brick_volume")
        elif self.layout == _lib.PAIRED:
            _lib.check(self.lib.diffus_pair_volume(_vp(self.vol), *self.dims, _vp(self.vol_k), self.stream()), "diffus_pair_volume")

    # -- the stages ---------------------------------------------------------------------------------------------
    def fwd(self):
        _lib.check(self.lib.diffus_render_fwd(*self.common, _vp(self.frame), None, _vp(self.ws), self.ws.numel(),
                                              self.stream()), "diffus_render_fwd")
        self._stamp += 1
        self._median_of = self._inputs_now()

    def bwd(self, stages=_lib.BWD_ALL):
        _lib.check(self.lib.diffus_render_bwd(*self.common_bwd, _vp(self.gframe), _vp(self.gvol_k), _vp(self.touched),
                                              _vp(self.gsrc if self._pose_grads else None),
                                              _vp(self.gdirs if self._pose_grads else None), stages, _vp(self.ws), self.ws.numel(),
                                              self.stream()), "diffus_render_bwd")

    def bwd_mse(self, stages=_lib.BWD_ALL):
        """Backward of loss_p = loss_scale * sum((frame_p - target_p)^2), straight from `self.frame`; `self.loss` gets loss_p."""
        _lib.check(self.lib.diffus_render_bwd_mse(*self.common_bwd, _vp(self.frame), _vp(self.target), self.loss_scale,
                                                  _vp(self.loss), _vp(self.gvol_k), _vp(self.touched),
                                                  _vp(self.gsrc if self._pose_grads else None),
                                                  _vp(self.gdirs if self._pose_grads else None), stages, _vp(self.ws),
                                                  self.ws.numel(), self.stream()),
                   "diffus_render_bwd_mse")

    def loss_and_grad(self):
        """loss_p = sum(frame_p^2), dL/dframe = 2 frame (one launch; the unfused form of step()'s loss)."""
        _lib.check(self.lib.diffus_loss_sumsq(_vp(self.frame), self.P, self.R * self.N1, _vp(self.loss),
                                              _vp(self.gframe), _vp(self.loss_ws), self.loss_ws.numel(), self.stream()),
                   "diffus_loss_sumsq")

    def zero_grad(self):
        """A fresh dense gradient every step: zero the caller's canonical (d0,d1,d2) tensor (sparse
        hand-back), or the bricked scratch (dense hand-back: the conversion overwrites every voxel)."""
        if self.gvol is not None and not self.persistent:
            (self.gvol if (self.touched is not None or not self.grad_bricked) else self.gvol_k).zero_()

    def finish_grad(self):
        """touched bricks of the scratch -> the canonical gradient; scratch back to all-zero."""
        if self.grad_bricked and self.gvol is not None and self.touched is not None:
            # mode STORE: the tensor was zeroed this step and every touched voxel is written once
            _lib.check(self.lib.diffus_gradbuf_flush(_vp(self.gvol_k), _vp(self.touched), *self.dims, _vp(self.gvol),
                                                     2 if self.persistent else 0, self.stream()), "diffus_gradbuf_flush")
        elif self.grad_bricked and self.gvol is not None:
            _lib.check(self.lib.diffus_unbrick_volume(_vp(self.gvol_k), *self.dims, _vp(self.gvol), 0, self.stream()),
                       "diffus_unbrick_volume")

    def forward(self):
        """[volume re-conversion +] forward frame into `self.frame`."""
        if self.learnable_volume:
            self.refresh_volume()
        self.fwd()

    def backward(self):
        """`self.gframe` (dL/dframe, written by the caller or by loss_and_grad) -> gvol, gsrc, gdirs.
        Must follow `forward()` on the same inputs: with start > 0 it reuses the per-pose median the forward left in
        this object's own workspace (DIFFUS_BWD_KEEP_MEDIAN) instead of launching the median kernel again."""
        self.zero_grad()
        self.bwd(_lib.BWD_ALL | (_lib.BWD_KEEP_MEDIAN if self._median_valid() else 0))
        self.finish_grad()

    def step_mse(self, stages: int = _lib.BWD_ALL, epilogue: bool = True):
        """One pass: frame, loss = loss_scale * sum((frame - target)^2) per pose and gvol/gsrc/gdirs in one call.
        epilogue=False (timing one kernel): no per-pose loss / d/dsource sums, whose blocks would otherwise ride in the
        scatter launch or, with stages = SCAN alone, be a launch of their own."""
        _lib.check(self.lib.diffus_render_step_mse(*self.common_bwd, _vp(self.target), self.loss_scale, _vp(self.frame),
                                                   _vp(self.loss if epilogue else None), _vp(self.gvol_k), _vp(self.touched),
                                                   _vp(self.gsrc if (epilogue and self._pose_grads) else None),
                                                   _vp(self.gdirs if self._pose_grads else None),
                                                   stages | (_lib.BWD_REPAIR_FRAME if (self.repair_frames and epilogue) else 0),
                                                   _vp(self.ws), self.ws.numel(), self.stream()),
                   "diffus_render_step_mse")

    def step(self):
        # (a forked stream for zero_grad beside the forward was measured: the fork/join events cost
        # more than the 10 us they hide -- 0.217 vs 0.202 ms/step -- so the step stays on one stream)
        self._stamp += 1
        if self.fused_loss and self.one_pass:
            if self.learnable_volume:
                self.refresh_volume()
            self.zero_grad()
            self.step_mse(_lib.BWD_ALL)      # computes its own median (start > 0): valid for these inputs afterwards
            self.finish_grad()
            self._median_of = self._inputs_now()
            return
        self.forward()
        if self.fused_loss:
            self.zero_grad()
            self.bwd_mse(_lib.BWD_ALL | (_lib.BWD_KEEP_MEDIAN if self._median_valid() else 0))
            self.finish_grad()
        else:
            if self.target is not None or self.loss_scale != 1.0:
                raise _lib.DiffusError("the unfused step() implements the sum-of-squares loss only (no target, scale 1)")
            self.loss_and_grad()
            self.backward()

    # -- hipGraph -------------------------------------------------------------------------------------------------
    def capture(self, what: str = "step", warmup: int = 2, repeat: int = 1):
        """Capture `what` ("step", "forward" or "backward") on a side stream; returns the torch.cuda.CUDAGraph.
        `repeat` > 1 captures that many consecutive calls into the one graph.

        When to capture (measured on MI355X / ROCm 7, tools/graph_gaps.py and tools/graph_batching.py): kernels of ONE
        graph -- like consecutive launches of one stream -- run back to back, but two graph LAUNCHES are ~8.6 us apart.
        A graph pays off when it replaces many launches or torch ops (a whole training iteration: 0.41 ms eager, 0.073 ms
        captured); for the bare `step()` -- two C-ABI calls, three kernels, ~15 us of host time -- issuing it eagerly is
        FASTER than one graph per step (63.8 against 68.6 us), and a graph of `repeat=8` steps equals the eager rate."""
        one = getattr(self, what)
        box_in = self.dirty_box          # what the captured conversion launch will cover
        if self.slice_only and box_in == _EMPTY_BOX and getattr(self, "_last_box", None) is not None:
            box_in = self._last_box      # capture() after a first eager iteration: that iteration's slice
        if self.slice_only and what in ("forward", "step"):
            # the warm-up calls below convert (and clear) the pending slice; the captured call must see it again
            inner = one

            def one():
                self._dirty_box = box_in
                inner()
        if repeat > 1:
            def fn():
                for _ in range(repeat):
                    one()
        else:
            fn = one
        if self._side is None:
            self._side = torch.cuda.Stream(self.dev)
        self._side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self._side):
            for _ in range(warmup):
                fn()
        self._side.synchronize()
        g = torch.cuda.CUDAGraph()
        keeps = what == "backward" and self._median_valid()      # the stages word is baked into the captured launch
        with torch.cuda.graph(g, stream=self._side):
            fn()
        self._graphs[what] = g
        self._graph_box[what] = box_in
        self._graph_keeps_median = getattr(self, "_graph_keeps_median", {})
        self._graph_keeps_median[what] = keeps
        return g

    def replay(self, what: str = "step"):
        """Replay the captured graph of `what`.  A "backward" graph captured with the forward's median baked in
        (DIFFUS_BWD_KEEP_MEDIAN, start > 0) is only valid for the inputs that median was computed from: after set_poses()
        or a volume edit without a forward in between it runs eagerly instead (same guard as the autograd path)."""
        if what not in self._graphs:
            raise _lib.DiffusError(f"replay({what!r}): nothing captured under that name (capture() first)")
        self._run(what)

    def _run(self, what: str):
        g = self._graphs.get(what)
        if g is not None and what == "backward" and getattr(self, "_graph_keeps_median", {}).get(what) \
                and not self._median_valid():
            g = None            # the captured launch would reuse a median of other inputs: recompute it eagerly instead
        if g is not None and what in ("forward", "step") and self.learnable_volume and self.layout != _lib.CANONICAL \
                and (self._full_once or self._graph_box.get(what) != self.dirty_box):
            g = None            # the captured conversion covers another part of the volume than the one that changed
        if g is not None:
            g.replay()
            if what in ("forward", "step"):     # the replayed launches leave the median of the current inputs
                self._median_of = self._inputs_now()
                self._stamp += 1
                if self.slice_only and self.learnable_volume:
                    self._dirty_box = _EMPTY_BOX    # the replayed conversion covered the pending slice
        else:
            if what in self._graphs:
                self.eager_fallbacks += 1
                if self.eager_fallbacks == 1:
                    import warnings
                    warnings.warn(f"CapturedStep.replay({what!r}): the captured launches do not fit the current inputs (another "
                                  "dirty box / a stale median); running eagerly -- see CapturedStep.eager_fallbacks", stacklevel=3)
            getattr(self, what)()

    # -- autograd ----------------------------------------------------------------------------------------------------
    def volume_with_slice(self, values: torch.Tensor, dim: int, index: int) -> torch.Tensor:
        """The step's volume with ONE slice replaced by `values` (e.g. the impedance an MLP predicts for the imaging
        plane -- `Z_vol = x.clone(); Z_vol[:, :, k] = Z_slice` of the reference's training notebook, cell 16 -- without
        cloning 64 MiB per iteration): written in place, returned as an autograd node whose backward hands `values`
        its slice of d/dvolume.  Pass the result to `render()`.  Canonical layout, or `learnable_volume=True`."""
        if self.layout != _lib.CANONICAL and not self.learnable_volume:
            raise _lib.DiffusError("volume_with_slice(): the converted copy must follow the volume: use layout='canonical' "
                                   "or learnable_volume=True")
        return _SliceIntoVolume.apply(self, values, int(dim), int(index))

    def _mark_slice(self, dim: int, index: int):
        """learnable_volume="slice": the next conversion rebuilds the records of this slice only -- or, when other slices
        have been written since the last conversion, of the box that holds all of them (nothing written is ever left stale;
        refresh_volume() empties the pending box)."""
        if self.slice_only:
            box = [(0, n) for n in self.dims]
            box[dim % 3] = (index % self.dims[dim % 3], index % self.dims[dim % 3] + 1)
            old = self._dirty_box
            if old is None and self.layout != _lib.CANONICAL:
                return                      # the whole volume is pending already
            if old is not None and old != _EMPTY_BOX:
                box = [(min(a, c), max(b, d)) for (a, b), (c, d) in zip(old, box)]
            self.dirty_box = tuple(box)

    def slice_view(self, dim: int, index: int) -> torch.Tensor:
        """The step's volume at `index` along `dim`, as a view (no gradient): somewhere for a producer to write a slice in
        place -- `model(x, scale, out=step.slice_view(2, k))` -- before `volume_with_slice()` / `mse_loss()` get it."""
        return self.vol.detach().select(int(dim), int(index))

    def _adopt(self, volume, sources, directions, what: str):
        """Copy arguments that are not the step's own tensors into them, in place; returns the three graph inputs."""
        v = self.vol if volume is None else volume
        s = self.src if sources is None else sources
        d = self.dirs if directions is None else directions
        with torch.no_grad():
            if v.data_ptr() != self.vol.data_ptr():
                if not self.learnable_volume and self.layout != _lib.CANONICAL:
                    raise _lib.DiffusError(f"{what}(): a volume other than the step's own needs learnable_volume=True "
                                           "(its converted copy must be rebuilt)")
                self.vol.copy_(v)
                self._full_once = True
                self._stamp += 1
            if s.data_ptr() != self.src.data_ptr():
                self.src.copy_(s.reshape(self.src.shape))
                self._stamp += 1
            if d.data_ptr() != self.dirs.data_ptr():
                self.dirs.copy_(d.reshape(self.dirs.shape))
                self._directions_changed()
                self._stamp += 1
        return v, s, d

    def set_target(self, target: Optional[torch.Tensor], loss_scale: Optional[float] = None):
        """The frames the fused loss compares with ((P,R,N1) float32, None: the frame's energy) and its scale -- e.g.
        1 / frame.numel() for torch's mse_loss.  Call before capture(): a captured graph holds the buffer's address
        (later calls with a tensor of the same shape copy into that buffer) and the scale's value."""
        if target is None:
            self.target = None
        else:
            t = target.detach().to(device=self.dev, dtype=torch.float32).reshape(self.P, self.R, self.N1)
            if self.target is not None and self.target.shape == t.shape:
                self.target.copy_(t)
            else:
                self.target = t.contiguous().clone()
        if loss_scale is not None:
            self.loss_scale = float(loss_scale)

    def mse_loss(self, volume: Optional[torch.Tensor] = None, sources: Optional[torch.Tensor] = None,
                 directions: Optional[torch.Tensor] = None, *, slice_values: Optional[torch.Tensor] = None,
                 slice_dim: int = 2, slice_index: int = 0) -> torch.Tensor:
        """sum over poses of loss_scale * sum((frame - target)^2) as ONE node of torch's autograd graph: its forward
        runs the one-pass step (diffus_render_step_mse: frame, losses and gradients from a single pass over the
        samples -- no forward launch, no loss kernels, nothing left for the backward but a scaling), `loss.backward()`
        hands d/dvolume, d/dsources, d/ddirections to the arguments that require grad.  The training loop of the
        reference's `[DEMO] Train MRI to Impedance MLP - GPU` cell 16 in one call:

            loss = step.mse_loss(slice_values=model(mri_slice, scale=1e6), slice_dim=2, slice_index=k)

        writes the predicted impedance into slice k of the step's volume (in place) and returns to `slice_values` that
        slice of d/dvolume (canonical layout, or learnable_volume=True).  `step.frame` holds the frame afterwards."""
        if not (self.fused_loss and self.one_pass):
            raise _lib.DiffusError("mse_loss() is the one-pass step: fused_loss and one_pass must be on")
        if slice_values is not None and self.layout != _lib.CANONICAL and not self.learnable_volume:
            raise _lib.DiffusError("mse_loss(slice_values=...): the converted copy must follow the volume: use "
                                   "layout='canonical' or learnable_volume=True")
        v, s, d = self._adopt(volume, sources, directions, "mse_loss")
        return _CapturedMSE.apply(self, v, s, d, slice_values, int(slice_dim), int(slice_index))

    def render(self, volume: Optional[torch.Tensor] = None, sources: Optional[torch.Tensor] = None,
               directions: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Forward frame (P,R,N1) as a node of torch's autograd graph: `loss.backward()` runs the step's backward and
        delivers d/dvolume, d/dsources, d/ddirections to whichever of the three arguments require grad.

        Arguments that are not the step's own tensors are copied into them in place (a 256^3 volume: 64 MiB, ~20 us on
        the device); a volume that changes between calls needs `learnable_volume=True` (its converted copy is then
        rebuilt inside the forward).  The returned frame aliases the step's frame buffer: it is valid until the next
        forward.  Uses the captured "forward" / "backward" graphs when `capture("forward")` / `capture("backward")`
        were called, eager launches otherwise (e.g. inside a caller's own torch.cuda.graph capture)."""
        v, s, d = self._adopt(volume, sources, directions, "render")
        return _CapturedRender.apply(self, v, s, d)
