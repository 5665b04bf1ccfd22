"""Gradient hand-back and loss utility on the GPU: diffus_gradbuf_flush in its three modes (the PERSISTENT mode
must leave exactly the dense gradient of the latest step in a tensor that is never memset) and diffus_loss_sumsq
(one launch, last-arriving block adds the partials: deterministic, counters back to zero)."""
import ctypes as C

import numpy as np
import pytest
import torch

from diffus_amd.phantom import phantom, pose_ring

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Step():
    """The captured-step API of the package."""
    from diffus_amd import CapturedStep
    return CapturedStep


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
def test_persistent_flush_equals_fresh_dense_gradient_every_step(Step, sampler):
    n, R, S = 64, 24, 90
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 12, R)
    groups = [[0, 1, 2], [5, 6], [2, 9, 10, 11], [3], [0, 1, 2], [7, 8]]      # fans move: bricks appear and disappear
    ref = None
    per = None
    for step, g in enumerate(groups * 2):
        s = torch.from_numpy(src[g]).cuda().contiguous()
        d = torch.from_numpy(dirs[g]).cuda().contiguous()
        hp_ref = Step(vol, s, d, S, 2e-3, sampler, persistent=False)
        hp_ref.step()
        hp = Step(vol, s, d, S, 2e-3, sampler, persistent=True)
        if per is not None:                    # carry the persistent tensor, scratch and flags across steps
            hp.gvol, hp.gvol_k, hp.touched = per
        assert hp.persistent
        hp.step()
        per = (hp.gvol, hp.gvol_k, hp.touched)
        torch.cuda.synchronize()
        # the same voxels are non-zero, nothing of earlier steps is left; values agree up to the order of the float atomics
        assert torch.equal(hp.gvol != 0, hp_ref.gvol != 0), (step, g)
        assert float((hp.gvol - hp_ref.gvol).abs().max()) <= 1e-5 * float(hp_ref.gvol.abs().max()), (step, g)
        assert float(hp.gvol.abs().max()) > 0
        assert torch.all(hp.gvol_k == 0)                              # the scratch is all-zero again
        fl = hp.touched.cpu().numpy()
        assert set(np.unique(fl)) <= {0, 2}
        assert torch.equal(hp.gsrc, hp_ref.gsrc) and torch.equal(hp.gdirs, hp_ref.gdirs)
        if ref is None:
            ref = hp_ref.gvol.clone()
    # a STORE flush on top of a persistent pair still works (stale bricks are cleared, flags return to 0)
    hp.zero_grad = lambda: None
    hp.persistent = False
    hp.step()
    torch.cuda.synchronize()
    assert torch.equal(hp.gvol != 0, hp_ref.gvol != 0) and torch.all(hp.touched == 0)
    assert float((hp.gvol - hp_ref.gvol).abs().max()) <= 1e-5 * float(hp_ref.gvol.abs().max())


def test_flush_modes_direct():
    from diffus_amd import _lib
    lib = _lib.load()
    d = (9, 10, 7)
    nb = lib.diffus_brick_count(*d)
    nf = lib.diffus_bricked_floats(*d)
    g = torch.Generator().manual_seed(0)
    bricked = torch.zeros(nf, device="cuda")
    touched = torch.zeros(nb, dtype=torch.int32, device="cuda")
    vol = torch.full(d, 5.0, device="cuda")
    dense = torch.randn(d, generator=g).cuda()
    assert lib.diffus_brick_volume(vp(dense), *d, vp(bricked), None) == 0
    touched[::3] = 1
    assert lib.diffus_gradbuf_flush(vp(bricked), vp(touched), *d, vp(vol), 7, None) == -1      # unknown mode
    assert lib.diffus_gradbuf_flush(vp(bricked), vp(touched), *d, vp(vol), 1, None) == 0       # accumulate
    torch.cuda.synchronize()
    changed = vol != 5.0
    assert torch.allclose(vol[changed], dense[changed] + 5.0) and 0 < int(changed.sum()) < vol.numel()
    assert torch.all(touched == 0)


@pytest.mark.parametrize("d", [(9, 10, 7), (8, 8, 8), (5, 3, 1), (64, 64, 64), (33, 70, 130)])
def test_flush_mode_dense_writes_every_voxel(d):
    """DIFFUS_FLUSH_DENSE (the drop-in autograd path's hand-back): the canonical tensor -- uninitialised before -- comes
    back with the touched bricks' values and ZEROS everywhere else, the scratch and the flags are all-zero afterwards;
    odd shapes (edge bricks, an odd dim 2) included."""
    from diffus_amd import _lib
    lib = _lib.load()
    nb = lib.diffus_brick_count(*d)
    nf = lib.diffus_bricked_floats(*d)
    g = torch.Generator().manual_seed(sum(d))
    dense = torch.randn(d, generator=g).cuda()
    bricked = torch.zeros(nf, device="cuda")
    assert lib.diffus_brick_volume(vp(dense), *d, vp(bricked), None) == 0
    pick = (torch.rand(nb, generator=g) < 0.3).cuda()
    stale = (~pick) & (torch.rand(nb, generator=g) < 0.2).cuda()          # flag 2: left by a PERSISTENT flush, scratch zero there
    touched = pick.to(torch.int32) + 2 * stale.to(torch.int32)
    keep = bricked.view(nb, 32) * pick[:, None]
    bricked.copy_(keep.reshape(-1))                                       # only the picked bricks hold values
    out = torch.full(d, float("nan"), device="cuda")
    assert lib.diffus_gradbuf_flush(vp(bricked), vp(touched), *d, vp(out), _lib.FLUSH_DENSE, None) == 0
    torch.cuda.synchronize()
    # expected: the dense tensor where its brick was picked, zero elsewhere
    want = torch.zeros(nf, device="cuda")
    assert lib.diffus_brick_volume(vp(dense), *d, vp(want), None) == 0
    want = (want.view(nb, 32) * pick[:, None]).reshape(-1).contiguous()
    back = torch.empty(d, device="cuda")
    assert lib.diffus_unbrick_volume(vp(want), *d, vp(back), 0, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and torch.equal(out, back)
    assert torch.all(bricked == 0) and torch.all(touched == 0)


@pytest.mark.parametrize("mode", ["store", "accumulate", "persistent"])
@pytest.mark.parametrize("d,shift", [((9, 10, 7), 0), ((8, 8, 8), 0), ((5, 3, 1), 0), ((64, 64, 64), 0), ((33, 70, 130), 0),
                                     ((33, 70, 130), 1), ((40, 36, 129), 0)])
def test_flush_sparse_modes_all_shapes(d, shift, mode):
    """The sparse modes of diffus_gradbuf_flush (a wave per 256 bricks, eight bricks per trip, 16-byte reads of the scratch,
    8-byte words of the canonical tensor when dim 2 is even and the tensor 8-byte aligned): edge bricks, odd dim 2, a brick
    count that is no multiple of 256, a canonical tensor at an odd float offset (`shift`), stale bricks (flag 2)."""
    from diffus_amd import _lib
    lib = _lib.load()
    nb = lib.diffus_brick_count(*d)
    nf = lib.diffus_bricked_floats(*d)
    g = torch.Generator().manual_seed(sum(d) + shift)
    dense = torch.randn(d, generator=g).cuda()
    bricked = torch.zeros(nf, device="cuda")
    assert lib.diffus_brick_volume(vp(dense), *d, vp(bricked), None) == 0
    pick = (torch.rand(nb, generator=g) < 0.3).cuda()
    stale = (~pick) & (torch.rand(nb, generator=g) < 0.2).cuda()
    touched = pick.to(torch.int32) + 2 * stale.to(torch.int32)
    bricked.copy_((bricked.view(nb, 32) * pick[:, None]).reshape(-1))
    # what the picked / the stale bricks cover, as canonical masks and values
    def canon(per_brick):
        b = per_brick.reshape(-1).contiguous()
        out = torch.empty(d, device="cuda")
        assert lib.diffus_unbrick_volume(vp(b), *d, vp(out), 0, None) == 0
        torch.cuda.synchronize()
        return out
    ones = torch.ones(nb, 32, device="cuda")
    in_pick, in_stale = canon(ones * pick[:, None]) != 0, canon(ones * stale[:, None]) != 0
    vals = canon(bricked.view(nb, 32).clone())
    store = torch.empty(int(torch.tensor(d).prod()) + shift, device="cuda")
    out = store[shift:].view(d)
    out.copy_(torch.full(d, 5.0))
    code = {"store": _lib.FLUSH_STORE, "accumulate": _lib.FLUSH_ACCUMULATE, "persistent": _lib.FLUSH_PERSISTENT}[mode]
    assert lib.diffus_gradbuf_flush(vp(bricked), vp(touched), *d, vp(out), code, None) == 0
    torch.cuda.synchronize()
    want = torch.full(d, 5.0, device="cuda")
    if mode == "accumulate":
        want[in_pick] += vals[in_pick]
    else:
        want[in_pick] = vals[in_pick]
        want[in_stale] = 0.0                                             # a stale brick: cleared in `out`
    assert torch.equal(out, want)
    assert torch.all(bricked == 0)
    assert torch.equal(touched, (2 * pick.to(torch.int32)) if mode == "persistent" else torch.zeros_like(touched))


@pytest.mark.parametrize("layout", ["bricked", "paired"])
@pytest.mark.parametrize("d", [(9, 10, 7), (8, 8, 8), (5, 3, 1), (64, 64, 64), (33, 70, 300), (12, 9, 130)])
def test_convert_volume_box_equals_a_full_conversion(d, layout):
    """diffus_convert_volume_box after a change confined to a box == converting the whole changed volume: slices along
    every axis (the reference's training loop rewrites one slice per step), corners, random boxes, the whole volume, an
    empty box; the paired records' repeated neighbours (next column, next depth) included."""
    from diffus_amd import _lib
    lib = _lib.load()
    code = {"bricked": _lib.BRICKED, "paired": _lib.PAIRED}[layout]
    nf = lib.diffus_bricked_floats(*d) if layout == "bricked" else lib.diffus_paired_floats(*d)
    full = lib.diffus_brick_volume if layout == "bricked" else lib.diffus_pair_volume
    g = torch.Generator().manual_seed(sum(d))
    rng = np.random.default_rng(sum(d))
    vol = torch.randn(d, generator=g).cuda()
    conv = torch.zeros(nf, device="cuda")
    assert full(vp(vol), *d, vp(conv), None) == 0
    boxes = [((0, d[0]), (0, d[1]), (0, d[2])), ((0, 0), (0, d[1]), (0, d[2]))]
    for ax in range(3):                                                  # one slice along each axis: first, last, a middle one
        for k in {0, d[ax] - 1, d[ax] // 2}:
            b = [(0, d[0]), (0, d[1]), (0, d[2])]
            b[ax] = (k, k + 1)
            boxes.append(tuple(b))
    for _ in range(6):
        lo = [int(rng.integers(0, n)) for n in d]
        boxes.append(tuple((l, int(rng.integers(l + 1, n + 1))) for l, n in zip(lo, d)))
    for (x0, x1), (y0, y1), (z0, z1) in boxes:
        vol[x0:x1, y0:y1, z0:z1] = torch.randn((x1 - x0, y1 - y0, z1 - z0), generator=g).cuda()
        assert lib.diffus_convert_volume_box(vp(vol), *d, code, vp(conv), x0, x1, y0, y1, z0, z1, None) == 0
        want = torch.zeros(nf, device="cuda")
        assert full(vp(vol), *d, vp(want), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(conv, want), ((x0, x1), (y0, y1), (z0, z1))


@pytest.mark.parametrize("layout", ["bricked", "paired"])
def test_learnable_volume_step_with_a_dirty_box(Step, layout):
    """A volume of which the caller rewrites ONE slice between steps: step() with `dirty_box` set (only that slice's
    records re-converted) gives what a step with the full conversion gives."""
    n, P, R, S = 48, 2, 16, 64
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, P, R)
    s, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
    part = Step(vol.clone(), s, dd, S, 1e-4, "trilinear", layout=layout, learnable_volume=True)
    whole = Step(vol.clone(), s, dd, S, 1e-4, "trilinear", layout=layout, learnable_volume=True)
    g = torch.Generator().manual_seed(3)
    for it in range(4):
        ax, k = it % 3, 20 + it
        sl = [slice(None)] * 3
        sl[ax] = slice(k, k + 1)
        new = (1.5e6 + 1e5 * torch.randn(vol[tuple(sl)].shape, generator=g)).cuda()
        box = [(0, n)] * 3
        box[ax] = (k, k + 1)
        for st in (part, whole):
            st.vol[tuple(sl)] = new
        part.dirty_box = tuple(box)
        part.step()
        whole.step()
        torch.cuda.synchronize()
        assert torch.equal(part.vol_k, whole.vol_k)
        assert torch.equal(part.frame, whole.frame) and torch.equal(part.loss, whole.loss)
        assert float((part.gvol - whole.gvol).abs().max()) <= 1e-5 * float(whole.gvol.abs().max())


@pytest.mark.parametrize("layout", ["bricked", "paired"])
@pytest.mark.parametrize("captured", [False, True])
def test_learnable_volume_slice_mode_tracks_the_written_slice(Step, layout, captured):
    """learnable_volume="slice" (the reference's training loop: the volume changes only through the slice an MLP predicts):
    mse_loss(slice_values=...) / volume_with_slice() mark the slice, the step re-converts that slice's records only --
    same converted volume, loss and slice gradient as the whole-volume conversion; a captured step whose baked slice is
    not the one written falls back to eager launches; a foreign volume passed in is converted whole."""
    n, P, R, S = 48, 2, 16, 64
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, P, R)
    s, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
    g = torch.Generator().manual_seed(5)
    tgt = (0.05 * torch.randn((P, R, S), generator=g)).cuda()
    part = Step(vol.clone(), s, dd, S, 1e-4, "trilinear", layout=layout, learnable_volume="slice", target=tgt)
    whole = Step(vol.clone(), s, dd, S, 1e-4, "trilinear", layout=layout, learnable_volume=True, target=tgt)
    plan = [(2, 24), (2, 24), (2, 25), (0, 20), (1, 30), (2, 24)]
    for it, (dim, k) in enumerate(plan):
        base = (1.5e6 + 1e5 * torch.randn(tuple(vol.select(dim, k).shape), generator=g)).cuda()
        grads = []
        for st in (part, whole):
            vals = base.clone().requires_grad_(True)
            loss = st.mse_loss(slice_values=vals, slice_dim=dim, slice_index=k)
            loss.backward()
            torch.cuda.synchronize()
            grads.append((loss.detach().clone(), vals.grad.clone()))
        assert torch.equal(part.vol_k, whole.vol_k), (it, dim, k)
        assert torch.equal(grads[0][0], grads[1][0])
        assert float((grads[0][1] - grads[1][1]).abs().max()) <= 1e-5 * float(grads[1][1].abs().max())
        if captured and it == 0:
            part.capture("step")                                   # bakes the conversion of slice (2, 24)
    # a volume that is not the step's own: copied in, converted whole
    other = (vol * 1.01).contiguous()
    for st in (part, whole):
        st.mse_loss(other)
    torch.cuda.synchronize()
    assert torch.equal(part.vol_k, whole.vol_k) and torch.equal(part.loss, whole.loss)


@pytest.mark.parametrize("layout", ["bricked", "paired"])
def test_slice_mode_two_writes_before_one_conversion(Step, layout):
    """learnable_volume="slice" with TWO slices written before one conversion -- volume_with_slice(k1), then
    volume_with_slice(k2), then render(); or volume_with_slice(k1) followed by mse_loss(slice_index=k2).  The pending box is
    the union of what was written (nothing stale), and it is empty again after the conversion."""
    n, P, R, S = 48, 2, 16, 64
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, P, R)
    s, dd = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
    g = torch.Generator().manual_seed(9)
    tgt = (0.05 * torch.randn((P, R, S), generator=g)).cuda()
    part = Step(vol.clone(), s, dd, S, 1e-4, "trilinear", layout=layout, learnable_volume="slice", target=tgt)
    whole = Step(vol.clone(), s, dd, S, 1e-4, "trilinear", layout=layout, learnable_volume=True, target=tgt)
    for (d1, k1), (d2, k2), how in [((2, 24), (2, 27), "render"), ((2, 23), (0, 20), "render"), ((1, 11), (2, 24), "mse")]:
        a = (1.5e6 + 1e5 * torch.randn(tuple(vol.select(d1, k1).shape), generator=g)).cuda()
        b = (1.5e6 + 1e5 * torch.randn(tuple(vol.select(d2, k2).shape), generator=g)).cuda()
        outs = []
        for st in (part, whole):
            st.volume_with_slice(a, d1, k1)
            if how == "render":
                v = st.volume_with_slice(b, d2, k2)
                outs.append(st.render(v).clone())
            else:
                outs.append(st.mse_loss(slice_values=b, slice_dim=d2, slice_index=k2).clone())
            torch.cuda.synchronize()
        assert torch.equal(part.vol, whole.vol)
        assert torch.equal(part.vol_k, whole.vol_k), (d1, k1, d2, k2, how)
        assert torch.equal(outs[0], outs[1])
        assert part.dirty_box == ((0, 0), (0, 0), (0, 0))
    part.dirty_box = [[0, n], [0, n], [3, 4]]                 # lists are normalised to the tuple form graphs are compared with
    assert part.dirty_box == ((0, n), (0, n), (3, 4))
    with pytest.raises(ValueError):
        part.dirty_box = ((0, n + 1), (0, n), (0, 1))


@pytest.mark.parametrize("P,n", [(1, 4), (3, 1000), (32, 256 * 512), (5, 131073)])
def test_loss_sumsq_single_launch(P, n):
    from diffus_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(P + n)
    frame = torch.randn(P, n, generator=g).cuda()
    loss = torch.full((P,), -1.0, device="cuda")
    gframe = torch.empty_like(frame)
    ws = torch.zeros(max(512 * P, 512), dtype=torch.uint8, device="cuda")
    busy = torch.randn(1 << 22, device="cuda")
    outs = []
    for it in range(6):                          # repeated calls: the arrival counters must come back to zero
        if it % 2:
            busy.mul_(1.0001)                     # uneven load beside it
        loss.fill_(-1.0)
        assert lib.diffus_loss_sumsq(vp(frame), P, n, vp(loss), vp(gframe), vp(ws), ws.numel(), None) == 0
        torch.cuda.synchronize()
        outs.append(loss.clone())
    ref = (frame.double() ** 2).sum(1)
    assert torch.allclose(outs[0].double(), ref, rtol=2e-6)
    assert all(torch.equal(o, outs[0]) for o in outs)            # deterministic
    assert torch.equal(gframe, 2 * frame)
    assert torch.all(ws[:4 * P].view(torch.int32) == 0)
    assert lib.diffus_loss_sumsq(vp(frame), P, n, vp(loss), None, vp(ws), 256 * P, None) == -4


def test_step_is_hipgraph_capturable_and_replays_on_new_inputs(Step):
    """The C-ABI never allocates or synchronises: a whole step (forward, loss, backward, persistent flush) is captured
    once and replayed after the poses and the volume were changed IN PLACE; results equal the eager step's."""
    n, P, R, S = 64, 3, 24, 1100                      # S > 1024: the segmented launches are captured too
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 8, R)
    s = torch.from_numpy(src[:P]).cuda().contiguous()
    d = (torch.from_numpy(dirs[:P]) * 0.05).cuda().contiguous()
    hp = Step(vol, s, d, S, 1e-3, "trilinear", layout="bricked")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(2):
            hp.step()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        hp.step()
    # new inputs, in place (the bricked copy of the volume is part of the caller's state here: rebuild it in place too)
    s.copy_(torch.from_numpy(src[4:4 + P]).cuda())
    vol.mul_(1.01)
    from diffus_amd import _lib
    _lib.check(hp.lib.diffus_brick_volume(vp(vol), *hp.dims, vp(hp.vol_k), None), "brick")
    g.replay()
    torch.cuda.synchronize()
    got = (hp.frame.clone(), hp.loss.clone(), hp.gvol.clone(), hp.gsrc.clone(), hp.gdirs.clone())
    ref = Step(vol, s, d, S, 1e-3, "trilinear", layout="bricked", persistent=False)
    ref.step()
    torch.cuda.synchronize()
    assert torch.equal(got[0], ref.frame) and torch.equal(got[1], ref.loss)
    assert torch.equal(got[3], ref.gsrc) and torch.equal(got[4], ref.gdirs)
    assert torch.equal(got[2] != 0, ref.gvol != 0)
    assert float((got[2] - ref.gvol).abs().max()) <= 1e-5 * float(ref.gvol.abs().max())
    assert float(ref.gvol.abs().max()) > 0 and float(ref.frame.abs().max()) > 0


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
def test_start_crop_backward_reuses_the_forwards_median(Step, sampler):
    """start > 0: the captured step's backward keeps the median of its own forward (DIFFUS_BWD_KEEP_MEDIAN) and routes the
    median's gradient in extra blocks of the scatter launch; same result as the self-contained backward of the autograd
    path (median recomputed, routing and d/dsource reduction as a launch of their own when there is no scatter)."""
    import diffus_amd as da
    n, P, R, S, start, alpha = 64, 5, 40, 150, 37, 2e-3
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, P, R)
    s = torch.from_numpy(src).cuda()
    d = torch.from_numpy(dirs).cuda()
    hp = Step(vol, s, d, S, alpha, sampler, start=start, persistent=False)
    hp.step()
    hp.step()                                           # a second pass: gmed was reset by the first one
    v = vol.clone().requires_grad_(True)
    sa = s.clone().requires_grad_(True)
    dd = d.clone().requires_grad_(True)
    f = da.render_poses(v, sa, dd, S, alpha, start=start, sampler=sampler, layout="paired")
    (f ** 2).sum().backward()
    torch.cuda.synchronize()
    # the same frame bit for bit -- except on rays with |echo| > 1 somewhere, which the FORWARD kernel evaluates again in float64 and
    # the one-pass step only on request (DESIGN fact 45): those agree to the float32 scan's noise there
    fd = f.detach()
    echo = fd.abs() * torch.exp(alpha * torch.arange(fd.shape[-1], device=fd.device, dtype=torch.float32))
    well = echo.amax(dim=-1) <= 1.0
    assert int(well.sum()) >= well.numel() - 8
    assert torch.equal(fd[well], hp.frame[well])
    if not bool(well.all()):
        peak = fd[~well].abs().amax(dim=-1, keepdim=True)
        assert float(((fd[~well] - hp.frame[~well]).abs() / peak).max()) <= 3e-4
    assert float(hp.gvol.abs().max()) > 0
    assert float((hp.gvol - v.grad).abs().max()) <= 2e-5 * float(v.grad.abs().max())
    if sampler == "trilinear":
        assert float((hp.gsrc - sa.grad).abs().max()) <= 1e-5 * float(sa.grad.abs().max())
        assert float((hp.gdirs - dd.grad).abs().max()) <= 1e-5 * float(dd.grad.abs().max())
    # pose-gradient-only backward (no scatter launch to carry the per-pose epilogue)
    if sampler == "trilinear":
        hq = Step(vol, s, d, S, alpha, sampler, start=start, want_gvol=False)
        hq.step()
        torch.cuda.synchronize()
        assert float((hq.gsrc - sa.grad).abs().max()) <= 1e-5 * float(sa.grad.abs().max())
        assert float((hq.gdirs - dd.grad).abs().max()) <= 1e-5 * float(dd.grad.abs().max())


def test_stale_median_is_not_reused_after_the_inputs_changed(Step):
    """forward(), then set_poses() / an in-place volume edit / refresh_volume(), then backward() must NOT route
    the gradient of the start-crop median through the median of the OLD inputs (DIFFUS_BWD_KEEP_MEDIAN): the backward
    recomputes it.  Checked against a fresh step on the new inputs; and a frame rendered before the change can no longer
    be back-propagated through render()."""
    n, P, R, S, start, alpha = 64, 4, 40, 150, 37, 2e-3
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 8, R)
    s0, d0 = torch.from_numpy(src[:P]).cuda(), torch.from_numpy(dirs[:P]).cuda()
    s1, d1 = torch.from_numpy(src[4:4 + P]).cuda(), torch.from_numpy(dirs[4:4 + P]).cuda()
    w = torch.linspace(0.5, 2.0, S - start, device="cuda")

    G = (torch.randn(P, R, S - start, generator=torch.Generator().manual_seed(3)) * w.cpu()).cuda()

    def reference(v, s, d):
        r = Step(v, s.clone(), d.clone(), S, alpha, "trilinear", start=start, persistent=False, fused_loss=False)
        r.forward()
        r.gframe.copy_(G)
        r.backward()
        torch.cuda.synchronize()
        return r

    for change in ("poses", "volume", "refresh"):
        v = vol.clone()
        hp = Step(v, s0.clone(), d0.clone(), S, alpha, "trilinear", start=start, persistent=False, fused_loss=False,
                  layout="canonical" if change == "volume" else "paired")
        hp.forward()
        assert hp._median_valid()
        if change == "poses":
            hp.set_poses(s1, d1)
        elif change == "volume":
            v[:, :, : n // 2].mul_(1.5)                      # canonical layout: the kernels read the caller's tensor
        else:
            v[:, :, : n // 2].mul_(1.5)
            hp.refresh_volume()
        assert not hp._median_valid()
        hp.gframe.copy_(G)
        hp.backward()                                        # no forward in between: the workspace median is the OLD inputs'
        torch.cuda.synchronize()
        ref = reference(v, s1 if change == "poses" else s0, d1 if change == "poses" else d0)
        assert float((hp.gsrc - ref.gsrc).abs().max()) <= 1e-5 * float(ref.gsrc.abs().max()), change
        assert float((hp.gdirs - ref.gdirs).abs().max()) <= 1e-5 * float(ref.gdirs.abs().max()), change
        assert float((hp.gvol - ref.gvol).abs().max()) <= 2e-5 * float(ref.gvol.abs().max()), change
    # the autograd node: a frame of the old poses cannot be back-propagated after set_poses
    hp = Step(vol, s0.clone().requires_grad_(True), d0.clone(), S, alpha, "trilinear", start=start)
    f = hp.render()
    hp.set_poses(s1, d1)
    with pytest.raises(RuntimeError):
        (f ** 2).sum().backward()


@pytest.mark.parametrize("S,start,sampler", [(512, 0, "trilinear"), (700, 30, "trilinear"), (1100, 0, "trilinear"),
                                             (150, 37, "nearest"), (96, 0, "trilinear")])
def test_fused_mse_backward_equals_loss_kernel_plus_backward(Step, S, start, sampler):
    """diffus_render_bwd_mse: loss = scale * sum((frame - target)^2) per pose and its backward, formed from the frame on
    the fly (single launch pair, per-pose loss summed by the call's closing blocks) -- against the unfused sequence
    (torch loss on the frame, diffus_render_bwd with the explicit dL/dframe).  Covers one wave per ray, two waves per ray
    (512 < N1 <= 1024), segmented rays (N1 > 1024) and the start crop."""
    n, P, R, alpha = 64, 3, 20, 1e-3
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 8, R)
    s = torch.from_numpy(src[:P]).cuda()
    d = (torch.from_numpy(dirs[:P]) * (0.08 if S > 600 else 1.0)).cuda().contiguous()
    g = torch.Generator().manual_seed(S)
    target = (torch.randn(P, R, S - start, generator=g) * 0.05).cuda()
    for tgt, scale in ((None, 1.0), (target, 0.37)):
        fused = Step(vol, s, d, S, alpha, sampler, start=start, persistent=False, target=tgt, loss_scale=scale,
                            one_pass=False)
        fused.step()
        # ... and the ONE-PASS step (diffus_render_step_mse: the frame too comes out of the adjoint-scan kernel)
        one = Step(vol, s, d, S, alpha, sampler, start=start, persistent=False, target=tgt, loss_scale=scale)
        one.frame.fill_(float("nan"))
        one.step()
        torch.cuda.synchronize()
        # same arithmetic; the scan may associate differently (chunk length, two waves per ray), hence rounding only
        # (each within 1e-5 of the float64 truth: tests/test_hip_parity.py)
        # The short steps of the long-ray cases put grazing interfaces on some rays (|echo| ~ 100: an ill-conditioned b/d,
        # DESIGN section 2); two float32 evaluations that associate the scan differently then agree to ~1e-4 only -- the
        # reference's own float32 is 1.4e-4 from its float64 there (golden G17) -- and so does the sum of squares
        fmax = float(fused.frame.abs().max())
        ill = fmax > 10.0
        assert float((one.frame - fused.frame).abs().max()) <= (2e-4 if ill else 2e-5) * fmax
        assert torch.allclose(one.loss, fused.loss, rtol=2e-4 if ill else 1e-5)
        assert float((one.gvol - fused.gvol).abs().max()) <= 1e-4 * float(fused.gvol.abs().max())
        assert torch.allclose(one.gsrc, fused.gsrc, rtol=1e-4, atol=1e-4 * float(fused.gsrc.abs().max()))
        assert torch.allclose(one.gdirs, fused.gdirs, rtol=1e-4, atol=1e-4 * float(fused.gdirs.abs().max()))
        ref = Step(vol, s, d, S, alpha, sampler, start=start, persistent=False, fused_loss=False)
        ref.fwd()
        diff = ref.frame if tgt is None else ref.frame - tgt
        ref.gframe.copy_(2 * scale * diff)
        ref.backward()
        torch.cuda.synchronize()
        want_loss = scale * (diff.double() ** 2).sum((1, 2))
        assert torch.equal(fused.frame, ref.frame)
        assert torch.allclose(fused.loss.double(), want_loss, rtol=2e-6), (fused.loss, want_loss)
        den = float(ref.gvol.abs().max())
        assert den > 0 and float((fused.gvol - ref.gvol).abs().max()) <= 2e-5 * den
        if sampler == "trilinear":
            assert float((fused.gsrc - ref.gsrc).abs().max()) <= 1e-5 * float(ref.gsrc.abs().max())
            assert float((fused.gdirs - ref.gdirs).abs().max()) <= 1e-5 * float(ref.gdirs.abs().max())
    # pose-gradient-only and loss-only calls
    if sampler == "trilinear":
        only = Step(vol, s, d, S, alpha, sampler, start=start, want_gvol=False)
        only.step()
        torch.cuda.synchronize()
        assert torch.allclose(only.loss.double(), (only.frame.double() ** 2).sum((1, 2)), rtol=2e-6)


@pytest.mark.parametrize("sampler", ["trilinear", "nearest"])
def test_canonical_volume_with_the_bricked_sparse_gradient(Step, sampler):
    """DIFFUS_GRAD_BRICKED: the kernels read the caller's canonical tensor in place, the gradient goes through the bricked
    scratch and the touched-brick hand-back (no memset of the dense tensor) -- same values as the scatter straight into
    the canonical tensor, step after step with fans that move and a volume slice that changes in place."""
    n, P, R, S, start = 64, 4, 24, 100, 9
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 12, R)
    s = torch.from_numpy(src[:P]).cuda().contiguous()
    d = torch.from_numpy(dirs[:P]).cuda().contiguous()
    a = Step(vol, s, d, S, 2e-3, sampler, start=start, layout="canonical")
    b = Step(vol, s, d, S, 2e-3, sampler, start=start, layout="canonical", bricked_grad=False)
    assert a.grad_bricked and a.persistent and a.touched is not None
    assert not b.grad_bricked and b.touched is None and b.gvol_k is b.gvol
    for it in range(4):
        lo = (it * 3) % 8
        for hp in (a, b):
            hp.set_poses(torch.from_numpy(src[lo:lo + P]).cuda(), torch.from_numpy(dirs[lo:lo + P]).cuda())
        vol[:, :, 30 + it].mul_(1.01)                       # both steps read the same tensor, in place
        a.step(); b.step()
        torch.cuda.synchronize()
        assert float(b.gvol.abs().max()) > 0
        # (the planar 64-bit tiles of the bricked path keep contributions the 32-bit tiles of the canonical one round to 0)
        assert not torch.any((b.gvol != 0) & (a.gvol == 0)), it
        assert float((a.gvol - b.gvol).abs().max()) <= 2e-5 * float(b.gvol.abs().max()), it
        assert torch.equal(a.frame, b.frame) and torch.equal(a.loss, b.loss)
        assert torch.equal(a.gsrc, b.gsrc) and torch.equal(a.gdirs, b.gdirs)
        assert torch.all(a.gvol_k == 0)


def test_a_graph_of_several_steps_leaves_what_one_step_leaves(Step):
    """capture(repeat=m): m consecutive steps in ONE hipGraph (two graph launches are ~8.6 us apart on this stack; DESIGN
    fact 22).  With the inputs unchanged every step rewrites the same results -- frame, per-pose losses and pose
    gradients bit for bit, the volume gradient up to the order of its float atomics, nothing of earlier steps left in
    the persistent tensor, scratch and flags consistent for the next replay."""
    n, R, S = 64, 40, 130
    vol = torch.from_numpy(phantom(n)).cuda()
    src, dirs = pose_ring(n, 5, R)
    s, d = torch.from_numpy(src).cuda(), torch.from_numpy(dirs).cuda()
    one = Step(vol, s, d, S, 2e-3, "trilinear")
    one.step()
    torch.cuda.synchronize()
    many = Step(vol, s, d, S, 2e-3, "trilinear")
    g = many.capture("step", repeat=3)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(many.frame, one.frame) and torch.equal(many.loss, one.loss)
    assert torch.equal(many.gsrc, one.gsrc) and torch.equal(many.gdirs, one.gdirs)
    assert torch.equal(many.gvol != 0, one.gvol != 0)
    assert float((many.gvol - one.gvol).abs().max()) <= 1e-5 * float(one.gvol.abs().max())
    assert torch.all(many.gvol_k == 0)
    with pytest.raises(AttributeError):
        many.capture("no_such_stage")
