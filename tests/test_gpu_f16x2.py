"""The f16x2 form of the dense layer (radargnn_amd/csrc/linear_dma.hip, FMT 1): every fp32 operand as two f16 terms after an
exact power-of-two pre-scale derived from a device-side bound, three matrix-pipe products per fp32 product.  Replaces ATen
addmm behind torch_geometric's Linear like the other dense kernels (gnn/gnn_models.py:137-178, gnn/mpnn_layers.py:64-74,89-90).

Tolerance: norm-wise against float64, max|a - b| <= 1e-6 max|b| -- the judge's mark for this kernel, an order inside the
1e-5 of the north_star -- next to the fp32 MFMA kernel on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    import radargnn_amd.gnn as gnn
    from radargnn_amd import ops
    return gnn, ops


def normwise(a, b) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def with_bound(t, slack=1.0):
    from radargnn_amd import ops
    t = t.cuda()
    ops.set_bound(t, ops.make_bound(t.abs().max() * slack))
    return t


SHAPES = [(3000, 224, 464, 224, False), (2305, 128, 272, 64, True), (4097, 64, 0, 132, False), (2560, 224, 0, 928, True),
          (192, 688, 0, 224, False)]


@pytest.mark.parametrize("m,k1,k2,n,relu", SHAPES)
@pytest.mark.parametrize("mag1,mag2", [(1.0, 1.0), (1e-6, 1e-6), (1e4, 1e4), (1e4, 1e-3), (3e-5, 70.0)])
def test_f16x2_form_is_as_accurate_as_fp32_mfma(rg, m, k1, k2, n, relu, mag1, mag2):
    """Adversarial magnitudes (whole tensors at 1e-6 ... 1e4, the two input blocks four to seven decades apart, columns spread
    over four decades): <= 1e-6 norm-wise at K = 688, never worse than the fp32 MFMA kernel's class; the launch really took the
    f16x2 form; max |out| is tracked exactly; the column statistics are those of the stored values."""
    _, ops = rg
    g = torch.Generator().manual_seed(m * 7 + n)
    a1 = torch.randn(m, k1, generator=g) * torch.logspace(-2, 2, k1).view(1, -1) * mag1
    a2 = torch.randn(m, k2, generator=g) * mag2 if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)
    b = torch.randn(n, generator=g) * mag1
    a = a1 if a2 is None else torch.cat([a1, a2], 1)
    exp = a.double() @ w.double().t() + b.double()
    if relu:
        exp = exp.clamp_min(0)
    kw = dict(relu=relu, want_stats=True)
    before = ops.COUNTERS.get("f16x2", 0)
    with ops.bound_tracking("cuda"):
        out, st = ops.linear(with_bound(a1), w.cuda(), b.cuda(), a2=None if a2 is None else with_bound(a2), **kw)
    assert ops.COUNTERS.get("f16x2", 0) == before + 1, "the launch did not take the f16x2 form"
    ops.USE_BF16X3 = False
    try:
        out_f32, _ = ops.linear(a1.cuda(), w.cuda(), b.cuda(), a2=None if a2 is None else a2.cuda(), **kw)
    finally:
        ops.USE_BF16X3 = True
    e16, e32 = normwise(out, exp), normwise(out_f32, exp)
    assert torch.isfinite(out).all()
    assert e16 < 1e-6 and e32 < 4e-6, (e16, e32)          # (the fp32 MFMA kernel is the comparator, not under test)
    assert e16 < 4 * e32 + 2e-7
    assert float(ops.bound_of(out).max()) == float(out.abs().max())
    s = [t_.cpu() for t_ in ops.stats_to_sums(st)[1:]]
    np.testing.assert_allclose(s[0], exp.sum(0), rtol=1e-4, atol=1e-5 * float(exp.abs().sum(0).max()))
    np.testing.assert_allclose(s[1], (exp * exp).sum(0), rtol=1e-4)


@pytest.mark.parametrize("slack,tol", [(2.0 ** 10, 1e-6), (2.0 ** 18, 1e-6), (2.0 ** 22, 4e-5), (2.0 ** 26, 1e-3)])
def test_f16x2_loose_bounds_degrade_gracefully(rg, slack, tol):
    """A bound far above the tensor's magnitude costs nothing up to 2^18 and then absolute accuracy 2^-40 of the BOUND per
    element -- never an overflow, never a non-finite value (the bounds BatchNorm hands on are loose by construction)."""
    _, ops = rg
    g = torch.Generator().manual_seed(5)
    m, k1, k2, n = 2000, 224, 464, 224
    a1, a2 = torch.randn(m, k1, generator=g), torch.randn(m, k2, generator=g)
    w, b = torch.randn(n, k1 + k2, generator=g) / 26.0, torch.randn(n, generator=g)
    exp = torch.cat([a1, a2], 1).double() @ w.double().t() + b.double()
    with ops.bound_tracking("cuda"):
        out = ops.linear(with_bound(a1, slack), w.cuda(), b.cuda(), a2=with_bound(a2, slack))
    assert torch.isfinite(out).all() and normwise(out, exp) < tol, normwise(out, exp)


def test_f16x2_row_subsets_and_affine_match_the_bf16x3_form(rg):
    """Row-subset launches with the BatchNorm-apply in the A path (how a conv layer's update runs): the f16x2 form agrees with
    the bf16x3 form to 1e-6 of the output's magnitude, writes the same rows and no others, and both halves of a split update
    share one bound word."""
    _, ops = rg
    g = torch.Generator().manual_seed(9)
    m, k1, k2, n = 6000, 224, 464, 224
    x, mm = torch.randn(m, k1, generator=g).cuda() * 3, torch.randn(m, k2, generator=g).cuda()
    w, b = (torch.randn(n, k1 + k2, generator=g) / 26.0).cuda(), torch.randn(n, generator=g).cuda()
    ss = torch.stack([torch.randn(k1, generator=g), torch.rand(k1, generator=g) + 0.5, torch.randn(k1, generator=g)]).cuda()   # mean_hi, g, t
    perm = torch.randperm(m, generator=g)
    rows_a, rows_b = perm[:3500].sort().values.int().cuda(), perm[3500:].sort().values.int().cuda()
    cnt_a, cnt_b = torch.tensor([3500]).cuda(), torch.tensor([2500]).cuda()

    def run(f16):
        out = torch.full((m, n), 7.0, device="cuda")
        with ops.bound_tracking("cuda"):
            if f16:
                x_, m_ = x.clone(), mm.clone()
                ops.set_bound(m_, ops.make_bound(mm.abs().max()))
                ss_ = ss.clone()
                ops.set_bound(ss_, ops.make_bound((x.abs().max() + ss[0].abs().max()) * ss[1].abs().max() + ss[2].abs().max()))
            else:
                x_, m_, ss_ = x, mm, ss
            ops.linear(x_, w, b, a2=m_, out=out, row_index=rows_a, m_dev=cnt_a, a1_affine=ss_)
            first = ops.bound_of(out)
            ops.linear(x_, w[:, :k1], b, out=out, row_index=rows_b, m_dev=cnt_b, a1_affine=ss_)
            assert ops.bound_of(out) is first or first is None
        return out

    before = ops.COUNTERS.get("f16x2", 0)
    o16 = run(True)
    assert ops.COUNTERS.get("f16x2", 0) == before + 2
    o3 = run(False)
    assert float(ops.bound_of(o16).max()) == float(o16.abs().max())
    assert normwise(o16, o3) < 1e-6


def test_detnet_forward_takes_the_f16x2_form_and_matches_the_bf16x3_model(rg):
    """DetNetBasic's inference forward tracks bounds by itself: the wide dense layers run in the f16x2 form, and logits / boxes
    agree with the bf16x3 model (RGNN_NO_F16X2) to 5e-6 norm-wise (three BatchNorm layers amplify the last-bit differences) -- both are compared with the float64 oracle elsewhere."""
    gnn, ops = rg
    from radargnn_amd import frames, synthetic
    cfg = gnn.GNNArchitectureConfig(5, 2, [224, 224, 64], [6], [16, 5], True, True, [32, 64, 128, 224], [4, 8, 16], "MPNNConv", False)
    torch.manual_seed(3)
    model = gnn.DetNetBasic(cfg).cuda()
    fr = [synthetic.radarscenes_frame(i) for i in range(3)]
    hp = frames.HotPath(model, frames.GraphSettings(algorithm="radius", k=0, r=1.0))
    batch = frames.FrameBatch.from_frames(fr)
    before = ops.COUNTERS.get("f16x2", 0)
    c16, b16, _ = hp(batch)
    took = ops.COUNTERS.get("f16x2", 0) - before
    assert took >= 8, took                       # 2 embedding layers + 3 x (source term, two update halves)
    ops.USE_F16X2 = False
    try:
        c3, b3, _ = hp(batch)
    finally:
        ops.USE_F16X2 = True
    assert normwise(c16, c3) < 5e-6 and normwise(b16, b3) < 5e-6, (normwise(c16, c3), normwise(b16, b3))


def _adversarial_matrix(m=20000, n=224, seed=11):
    """Columns of a layer output as an attacker would shape them: a handful of rows with entries of 1e6 in some columns, columns
    that are constant, constant up to one ulp, and constant except for ONE row, beside ordinary columns."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, n, generator=g)
    x[torch.randint(0, m, (12,), generator=g), :40] = 1e6 * torch.randn(12, 40, generator=g)     # outliers
    x[:, 50] = 0.5                                                                               # constant
    x[:, 51] = 5.0
    x[::2, 51] = float(np.nextafter(np.float32(5.0), np.float32(6.0)))                            # constant up to one ulp
    x[:, 52] = -3.0
    x[777, 52] = 1e6                                                                              # constant except one row
    x[:, 53] = 0.0                                                                                # all zero
    return x


@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_table_bound_is_valid_and_tight_on_adversarial_columns(rg, training):
    """The bound BatchNorm-finalize attaches to its apply table (what the f16x2 dense form scales its A1 operand by) must (1) HOLD:
    no normalised value exceeds it -- an f16 overflow otherwise -- and (2) in training mode stay within 2^12 of the largest
    normalised value even with 1e6 outliers beside constant columns (the any-table bound |g| (B + |mean|) + |t|
    sits 2^28 above the data there; csrc/norm.hip bn_bound).  Eval mode (running statistics: nothing ties the table to the data)
    keeps the any-table bound: checked for validity only."""
    _, ops = rg
    x = _adversarial_matrix().cuda()
    m, n = x.shape
    w = torch.eye(n, device="cuda")
    gamma = (torch.rand(n, generator=torch.Generator().manual_seed(1)) + 0.5).cuda()
    beta = torch.randn(n, generator=torch.Generator().manual_seed(2)).cuda()
    rm, rv = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda")
    with ops.bound_tracking("cuda"):
        xin = x.clone()
        ops.set_bound(xin, ops.make_bound(x.abs().max()))
        h, st = ops.linear(xin, w, None, want_stats=True)          # (identity layer: h = x up to the f16x2 rounding; the statistics are h's)
        ss = ops.batchnorm_finalize(st, m, n, gamma, beta, rm, rv, None, training, 0.1, 1e-5, in_bound=ops.bound_of(h))
        bound = float(ops.bound_of(ss).max())
        y = ops.scale_shift_act(h, ss, relu=False)
    ymax = float(y.abs().max())
    assert np.isfinite(bound) and ymax <= bound, (ymax, bound)
    # float64 reference of the normalised values (training: batch statistics)
    xd = h.double()
    mean, var = (xd.mean(0), xd.var(0, unbiased=False)) if training else (torch.zeros(n, dtype=torch.float64, device="cuda"),
                                                                            torch.ones(n, dtype=torch.float64, device="cuda"))
    ref = (xd - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    if training:
        assert bound <= 2.0 ** 12 * ymax, (bound, ymax)
        assert bound <= float(gamma.max()) * np.sqrt(m) * 1.01 + float(beta.abs().max()) + 2.0 ** -18 * 316 * 2.1e6 * 1.6


def test_hot_path_with_outliers_and_dead_channels_stays_inside_1e5(rg):
    """a 64-frame batch in which one frame carries rcs / velocity outliers of 1e6, every point of the batch
    has the same time stamp (a constant input column) and every conv layer has output channels that are dead (zero weight row,
    constant bias: var = 0, g = gamma / sqrt(eps)).  With the any-table bound the BatchNorm tables' bounds sit 2^25 - 2^28 above
    the normalised activations and the f16x2 dense layers lose their low terms; with the batch-statistics bound (norm.hip
    bn_bound) the model stays inside 1e-5 of the float64 oracle, train mode (the reference's regime) -- graph replayed too."""
    gnn, ops = rg
    import bench
    from oracle import gnn_hoisted as GH
    from radargnn_amd import frames as fr, synthetic
    frames = [synthetic.radarscenes_frame(300 + i) for i in range(64)]
    rng = np.random.default_rng(5)
    bad = frames[17]
    idx = rng.choice(bad.n, 24, replace=False)
    rcs, V = bad.rcs.copy(), bad.V.copy()
    rcs[idx[:12], 0] = 1e6 * rng.choice([-1.0, 1.0], 12)
    V[idx[12:], :] = (1e6 * rng.standard_normal((12, 2))).astype(np.float32).astype(np.float64)
    frames[17] = synthetic.RadarFrame(bad.X, V, rcs, bad.timestamp)
    frames = [synthetic.RadarFrame(f.X, f.V, f.rcs, np.zeros_like(f.timestamp)) for f in frames]      # time_index == 0 everywhere
    cfg = bench.c2_settings()
    model = bench.c2_model(seed=4)
    with torch.no_grad():
        for l, conv in enumerate(model.convs):
            lin = conv.post_mlp[0]
            for j in (3, 40 + l):
                lin.weight[j].zero_()
                lin.bias[j] = 0.25 * (j + 1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    batch = fr.FrameBatch.from_frames(frames)
    before = ops.COUNTERS.get("f16x2", 0)
    hot = fr.HotPath(model, cfg, use_hip_graphs=True)
    for _ in range(4):
        cls, bb, g = hot(batch)
    g.check()
    assert ops.COUNTERS.get("f16x2", 0) - before >= 10, "the dense layers did not take the f16x2 form"
    c64, b64 = GH.det_net_basic_hoisted(g.x, g.edge_index, g.edge_attr, sd, device="cuda")
    ec, eb = normwise(cls, c64), normwise(bb, b64)
    from conftest import record_parity
    record_parity("adversarial batch (64 x 3000, r = 1 m: 1e6 outliers in one frame, constant time index, dead channels in every conv layer; train mode, HIP graph)",
                  logits=ec, boxes=eb)
    assert torch.isfinite(cls).all() and torch.isfinite(bb).all()
    assert ec < 1e-5 and eb < 1e-5, (ec, eb)
    # the SAME check on the ordinary nodes only: the outliers dominate max|b|, so the norm over all nodes says little about the
    # 191 976 nodes without them -- per-frame norms over the frames the outliers do not touch (they share BatchNorm statistics
    # with the outlier frame, nothing else)
    ptr = np.concatenate([[0], np.cumsum([f.n for f in frames])])
    worst = 0.0
    for f in (0, 16, 18, 63):
        sl = slice(int(ptr[f]), int(ptr[f + 1]))
        worst = max(worst, normwise(cls[sl], c64[sl]), normwise(bb[sl], b64[sl]))
    record_parity("adversarial batch: worst per-frame norm over frames without outliers", worst=worst)
    assert worst < 1e-5, worst          # (any-table bound, r05 library: 5.3e-4 here)
