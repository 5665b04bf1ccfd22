"""SURVEY §8f row 4 on the GPU: the fused impedance MLP (forward + backward), the brain mask, the z-score
statistics and compute_impedance_volume, against the reference's own outputs (golden G15), the NumPy oracle and
torch autograd of the same network."""
import numpy as np
import pytest
import torch

from conftest import load_golden, maxnorm_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da():
    import diffus_amd
    diffus_amd._lib.load()
    return diffus_amd


def _model(da, g):
    m = da.ImpedanceEstimator(1)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")})   # the reference's keys
    return m


def test_mlp_forward_backward_golden(da):
    g = load_golden("g15_impedance")
    m = _model(da, g).cuda()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = m(x)
    assert y.shape == x.shape and y.is_cuda
    assert maxnorm_rel(y.detach().cpu().numpy(), g["y"]) < 5e-6
    (y * torch.from_numpy(g["up"]).cuda()).sum().backward()
    assert maxnorm_rel(x.grad.cpu().numpy(), g["gx"]) < 1e-5
    for k, p in m.named_parameters():
        assert maxnorm_rel(p.grad.cpu().numpy(), g["g_" + k]) < 2e-5, k


@pytest.mark.parametrize("n", [1, 31, 32, 33, 63, 64, 65, 1000, 70001])
def test_mlp_sizes_masks_and_affine_vs_oracle(da, n):
    from oracle import impedance as oi
    g = load_golden("g15_impedance")
    params = oi.pack({k[3:]: g[k] for k in g.files if k.startswith("sd_")})
    rng = np.random.default_rng(n)
    x = rng.normal(120, 40, size=n).astype(np.float32)
    mask = rng.uniform(size=n) < 0.7
    if n > 200:
        mask[64:192] = False                 # whole 64-voxel groups of air: the skip path
    shift, div, scale, fill = np.float32(118.5), np.float32(37.25), 1e6, 400.0
    lib = da._lib.load()
    xd, md = torch.from_numpy(x).cuda(), torch.from_numpy(mask).cuda()
    pd = torch.from_numpy(params).cuda()
    y = torch.full((n,), -7.0, device="cuda")
    rc = lib.diffus_mlp_fwd(xd.data_ptr(), md.data_ptr(), n, pd.data_ptr(), float(shift), float(div), scale, fill,
                            y.data_ptr(), 1, None)
    assert rc == 0
    ref = np.where(mask, oi.mlp_forward((x - shift) / div, params, np.float64) * scale, fill)
    assert maxnorm_rel(y.cpu().numpy(), ref) < 5e-6
    assert np.all(y.cpu().numpy()[~mask] == fill)
    # backward: masked voxels carry no gradient; result is deterministic (fixed-order reductions)
    gy = rng.normal(size=n).astype(np.float32)
    gyd = torch.from_numpy(gy).cuda()
    ws = torch.empty(lib.diffus_mlp_workspace_bytes(), dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        gp = torch.full((1153,), 3.0, device="cuda")
        gx = torch.full((n,), 3.0, device="cuda")
        rc = lib.diffus_mlp_bwd(xd.data_ptr(), md.data_ptr(), n, pd.data_ptr(), float(shift), float(div), scale,
                                gyd.data_ptr(), 1, gp.data_ptr(), gx.data_ptr(), ws.data_ptr(), ws.numel(), None)
        assert rc == 0
        outs.append((gp.cpu().numpy(), gx.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    gp_ref, gx_ref = oi.mlp_backward(((x - shift) / div)[mask], params, gy[mask] * scale)
    assert maxnorm_rel(outs[0][0], gp_ref) < 2e-5
    gxf = np.zeros(n)
    gxf[mask] = gx_ref / div
    assert maxnorm_rel(outs[0][1], gxf) < 1e-5 or np.abs(gxf).max() == 0


def test_mlp_matches_torch_autograd_of_the_same_network(da):
    torch.manual_seed(3)
    m = da.ImpedanceEstimator(1).cuda()
    x = torch.randn(4, 33, 17, device="cuda") * 2           # any shape, like the notebook's ImpedanceLearner.forward
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    up = torch.randn_like(y)
    (y * up).sum().backward()
    xt = x.clone().requires_grad_(True)
    yt = m.model(xt.reshape(-1, 1)).reshape(x.shape)          # plain torch layers with the same parameters
    gt = torch.autograd.grad((yt * up).sum(), [xt] + list(m.parameters()))
    assert maxnorm_rel(y.detach().cpu().numpy(), yt.detach().cpu().numpy()) < 5e-6
    assert maxnorm_rel(xr.grad.cpu().numpy(), gt[0].cpu().numpy()) < 1e-5
    for p, gref in zip(m.parameters(), gt[1:]):
        assert maxnorm_rel(p.grad.cpu().numpy(), gref.cpu().numpy()) < 2e-5


def test_brain_mask_stats_and_impedance_volume_golden(da):
    from oracle import impedance as oi
    g = load_golden("g15_impedance")
    m = _model(da, g)
    mri = torch.from_numpy(g["mri"])
    for tag in ("t50", "t120"):
        thr = float(g[tag + "_thr"])
        mask = da.create_brain_mask(g["mri"], thr)
        assert mask.dtype == torch.bool and np.array_equal(mask.numpy(), g[tag + "_mask"])       # bit-exact
        mean, std, cnt = da.masked_stats(mri, mask)
        mo, so = oi.masked_stats(g["mri"], g[tag + "_mask"])
        assert cnt == int(g[tag + "_mask"].sum()) and abs(mean - mo) < 1e-9 * abs(mo) and abs(std - so) < 1e-9 * so
        vn = da.zscore_normalize(mri, mask)
        assert maxnorm_rel(vn.numpy(), g[tag + "_vnorm"]) < 1e-6
        Z = da.ImpedanceEstimator.compute_impedance_volume(mri, m, thr)
        assert Z.shape == mri.shape and Z.dtype == torch.float32
        assert maxnorm_rel(Z.numpy(), g[tag + "_Z"]) < 1e-5
        assert torch.all(Z[~mask] == 400.0)
    for it in (0, 1, 3):                                       # other iteration counts vs the oracle's morphology
        mk = da.create_brain_mask(mri.cuda(), 50, iterations=it)
        assert mk.is_cuda and np.array_equal(mk.cpu().numpy(), oi.create_brain_mask(g["mri"], 50, it))


def test_train_model_and_render_through_the_estimator(da):
    """The training loop of `[DEMO] Train MRI to Impedance MLP - GPU` cell 16 with every stage in HIP kernels:
    MRI slice -> fused MLP -> volume -> renderer -> loss; gradients reach the MLP's parameters."""
    torch.manual_seed(0)
    X = torch.linspace(-2, 2, 256).reshape(-1, 1)
    yv = 1.5 + 0.2 * X + 0.1 * torch.sin(3 * X)
    m = da.ImpedanceEstimator.train_model(X, yv, epochs=300, lr=1e-2)
    assert float(((m(X) - yv) ** 2).mean().detach()) < 2e-3
    from diffus_amd.phantom import pose_ring
    n = 32
    mri = torch.rand(n, n, n, device="cuda") * 2 - 1
    m = m.cuda()
    src, dirs = pose_ring(n, 2, 16)
    Z = m(mri) * 1e6
    f = da.render_poses(Z, torch.from_numpy(src), torch.from_numpy(dirs), 40, 1e-3, sampler="trilinear")
    (f ** 2).sum().backward()
    gs = [p.grad for p in m.parameters()]
    assert all(g_ is not None and torch.isfinite(g_).all() for g_ in gs) and any(float(g_.abs().max()) > 0 for g_ in gs)


def test_mlp_writes_into_a_slice_and_reads_a_strided_gradient(da):
    """`model(x, scale, out=slice of a volume)`: the prediction lands in place (y_stride = the volume's last dimension), the
    upstream gradient -- the same slice of d/dvolume, a strided view -- is read in place (gy_stride); both against the
    contiguous path, bit for bit.  The six parameters become views of one flat buffer (no torch.cat per forward), values,
    names and state_dict unchanged."""
    torch.manual_seed(3)
    net = da.ImpedanceEstimator().cuda()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(24, 40, device="cuda")
    vol = torch.full((24, 40, 7), -1.0, device="cuda")
    gvol = torch.randn(24, 40, 7, device="cuda")
    y0 = net(x, scale=1e3)                                       # contiguous path (also flattens the parameters)
    ps = net._params()
    assert all(p.untyped_storage().data_ptr() == ps[0].untyped_storage().data_ptr() for p in ps)
    assert all(torch.equal(before[k], v) for k, v in net.state_dict().items()) and list(before) == list(net.state_dict())
    (y0 * gvol[:, :, 3].contiguous()).sum().backward()
    g0 = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    y1 = net(x, scale=1e3, out=vol[:, :, 3])
    assert y1.data_ptr() == vol[:, :, 3].data_ptr() and y1.stride() == vol[:, :, 3].stride()
    assert torch.equal(vol[:, :, 3], y0) and torch.all(vol[:, :, 2] == -1) and torch.all(vol[:, :, 4] == -1)
    (y1 * gvol[:, :, 3]).sum().backward()                        # mul's backward hands a strided gradient: read in place
    for a, b in zip(g0, [p.grad for p in net.parameters()]):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        net(x, out=torch.empty(24, 41, device="cuda"))
    with pytest.raises(ValueError):
        net(x, out=vol[:, 1, :40].t())                           # not a uniform stride over 24 x 40 elements
    # a Module.to() round trip gives every parameter its own storage again; the next forward re-flattens, same values
    net = net.cpu().cuda()
    assert torch.equal(net(x, scale=1e3), y0)
