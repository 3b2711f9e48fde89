"""HIP GNN path (through the C ABI) against the CPU oracle (oracle/gnn_oracle.py) and the reference's
known-answer tests (test/test_gnn.py).

Tolerance (BASELINE.json north_star: 1e-5 relative on float features / logits), written as the norm-wise bound
SURVEY.md section 7.3 derives:   max|a - b| <= 1e-5 * max|b|   per output tensor, b = float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import gnn_oracle as G
from oracle import graph_oracle as go
from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def rg():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    import radargnn_amd.gnn as gnn
    from radargnn_amd import ops
    return gnn, ops


def normwise(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def cpu_sd(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def set_ones(seq, Linear, value=1.0):
    for layer in seq:
        if isinstance(layer, Linear):
            layer.weight = torch.nn.Parameter(torch.ones_like(layer.weight) * value)
            layer.bias = torch.nn.Parameter(torch.zeros_like(layer.bias))


# ------------------------------------------------------------------------------------ dense layer
@pytest.mark.parametrize("m,k1,k2,n", [(1, 2, 0, 5), (7, 5, 0, 32), (300, 32, 0, 64), (1000, 224, 464, 224),
                                       (513, 128, 272, 64), (257, 64, 0, 6), (129, 33, 7, 130), (4096, 224, 0, 928)])
def test_linear_matches_fp64(rg, m, k1, k2, n):
    _, ops = rg
    g = torch.Generator().manual_seed(m + n)
    a1 = torch.randn(m, k1, generator=g)
    a2 = torch.randn(m, k2, generator=g) if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)
    b = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    a = a1 if a2 is None else torch.cat([a1, a2], 1)
    exp = a.double() @ w.double().t() + b.double()
    out, stats = ops.linear(a1.cuda(), w.cuda(), b.cuda(), a2=None if a2 is None else a2.cuda(), want_stats=True)
    assert normwise(out, exp) < 2e-6
    s = [t_.cpu() for t_ in ops.stats_to_sums(stats)[1:]]        # panels: {count, pivot, s1, s2} -> sum, sum of squares
    np.testing.assert_allclose(s[0], exp.sum(0), rtol=1e-4, atol=1e-3 * max(1.0, float(exp.abs().sum(0).max())) * 1e-2)
    np.testing.assert_allclose(s[1], (exp * exp).sum(0), rtol=1e-4)
    out2 = ops.linear(a1.cuda(), w.cuda(), b.cuda(), a2=None if a2 is None else a2.cuda(), relu=True)
    assert normwise(out2, exp.clamp_min(0)) < 2e-6
    out3 = ops.linear(a1.cuda(), w.cuda(), b.cuda(), a2=None if a2 is None else a2.cuda(), residual=res.cuda())
    assert normwise(out3, exp + res.double()) < 2e-6


@pytest.mark.parametrize("m,k1,k2,n,relu", [(2048, 32, 0, 32, False), (3000, 224, 464, 224, False), (5000, 36, 0, 68, True),
                                            (2305, 128, 272, 64, True), (4097, 64, 0, 132, False), (2560, 224, 0, 928, True)])
def test_linear_bf16x3_path_is_as_accurate_as_fp32_mfma(rg, m, k1, k2, n, relu):
    """Dense layer on the bf16 matrix pipe (three bf16 terms per fp32 operand, six products, fp32 accumulate; linear.hip
    k_linear_x3) against float64, next to the fp32 MFMA kernel on the same inputs: same error class (<= 2e-6 norm-wise),
    including partial row tiles (m % 256), a partial last k-step (K % 32), [A1|A2] inputs and the column statistics."""
    _, ops = rg
    g = torch.Generator().manual_seed(m * 7 + n)
    a1 = torch.randn(m, k1, generator=g) * torch.logspace(-2, 2, k1).view(1, -1)     # wide dynamic range per column
    a2 = torch.randn(m, k2, generator=g) if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)
    b = torch.randn(n, generator=g)
    a = a1 if a2 is None else torch.cat([a1, a2], 1)
    exp = a.double() @ w.double().t() + b.double()
    if relu:
        exp = exp.clamp_min(0)
    args = (a1.cuda(), w.cuda(), b.cuda())
    kw = dict(a2=None if a2 is None else a2.cuda(), relu=relu, want_stats=True)
    assert ops.USE_BF16X3 and m >= ops.BF16X3_MIN_ROWS
    min_cols, ops.BF16X3_MIN_COLS = ops.BF16X3_MIN_COLS, 0       # (narrow layers default to the fp32 kernel: force the x3 instances)
    try:
        out_x3, st_x3 = ops.linear(*args, **kw)
    finally:
        ops.BF16X3_MIN_COLS = min_cols
    ops.USE_BF16X3 = False
    try:
        out_f32, st_f32 = ops.linear(*args, **kw)
    finally:
        ops.USE_BF16X3 = True
    e_x3, e_f32 = normwise(out_x3, exp), normwise(out_f32, exp)
    assert e_x3 < 2e-6 and e_f32 < 2e-6, (e_x3, e_f32)
    assert e_x3 < 4 * e_f32 + 2e-7                              # not a weaker path: same error class as exact-fp32 products
    s = [t_.cpu() for t_ in ops.stats_to_sums(st_x3)[1:]]
    np.testing.assert_allclose(s[0], exp.sum(0), rtol=1e-4, atol=1e-5 * float(exp.abs().sum(0).max()))
    np.testing.assert_allclose(s[1], (exp * exp).sum(0), rtol=1e-4)
    assert st_x3.shape == st_f32.shape


@pytest.mark.parametrize("m,sub,k1,k2,n,relu", [(5000, 2777, 224, 464, 224, False), (4096, 4096, 224, 0, 224, False),
                                                 (3000, 0, 64, 0, 96, True), (9000, 300, 32, 32, 272, True),
                                                 (2500, 1201, 36, 0, 68, False)])
def test_linear_bf16x3_row_subset(rg, request, m, sub, k1, k2, n, relu):
    """Row-subset launches of the bf16x3 kernel (how MPNNConv updates the nodes with / without incoming edges): rows are
    gathered through an index list whose length lives on the device, results are scattered to the same rows, every other
    row of ``out`` stays untouched and the column statistics cover exactly the subset."""
    _, ops = rg
    g = torch.Generator().manual_seed(m + sub)
    a1 = torch.randn(m, k1, generator=g)
    a2 = torch.randn(m, k2, generator=g) if k2 else None
    w = torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)
    b = torch.randn(n, generator=g)
    rows = torch.randperm(m, generator=g)[:sub].sort().values
    lst = torch.full((m,), -7, dtype=torch.int32)               # entries past the count must never be read as rows
    lst[:sub] = rows.to(torch.int32)
    cnt = torch.tensor([sub], dtype=torch.int64)
    a = a1 if a2 is None else torch.cat([a1, a2], 1)
    exp = a[rows].double() @ w.double().t() + b.double()
    if relu:
        exp = exp.clamp_min(0)
    sentinel = 12345.0
    min_cols, ops.BF16X3_MIN_COLS = ops.BF16X3_MIN_COLS, 0     # (narrow layers default to the fp32 kernel: force the x3 instances)
    request.addfinalizer(lambda: setattr(ops, "BF16X3_MIN_COLS", min_cols))
    for x3 in (True, False):
        ops.USE_BF16X3 = x3
        try:
            out = torch.full((m, n), sentinel, dtype=torch.float32).cuda()
            panels = max(ops.stat_panels(m), 1)
            st = torch.zeros((panels, ops.STAT_ROWS, n), dtype=torch.float32).cuda()
            ops.linear(a1.cuda(), w.cuda(), b.cuda(), a2=None if a2 is None else a2.cuda(), relu=relu, out=out,
                       row_index=lst.cuda(), m_dev=cnt.cuda(), stats_out=st)
        finally:
            ops.USE_BF16X3 = True
        got = out.cpu()
        mask = torch.ones(m, dtype=torch.bool)
        mask[rows] = False
        assert torch.all(got[mask] == sentinel)
        if sub:
            assert normwise(got[rows], exp) < 2e-6
            cnt_, s1_, s2_ = (t_.cpu() for t_ in ops.stats_to_sums(st))      # panels hold {count, mean, M2}
            assert torch.all(cnt_ == sub)
            np.testing.assert_allclose(s1_, exp.sum(0), rtol=1e-4, atol=1e-5 * float(exp.abs().sum(0).max()) + 1e-6)
            np.testing.assert_allclose(s2_, (exp * exp).sum(0), rtol=1e-4, atol=1e-6)
        else:
            assert float(st.abs().max()) == 0.0


@pytest.mark.parametrize("m,k,n,relu,bias", [(5000, 5, 32, True, True), (8000, 2, 4, True, True), (4096, 8, 64, False, False),
                                             (6001, 3, 6, False, True), (7000, 1, 1, True, True)])
def test_linear_tiny_reduction_kernel(rg, m, k, n, relu, bias):
    """K <= 8 layers with many rows (first Linear of the node / edge embeddings) run on k_linear_tiny: an exact fp32 FMA
    chain in k order, checked against float64 and -- on integer data -- for exactness."""
    _, ops = rg
    g = torch.Generator().manual_seed(m + k + n)
    a = torch.randn(m, k, generator=g)
    w = torch.randn(n, k, generator=g)
    b = torch.randn(n, generator=g) if bias else None
    exp = a.double() @ w.double().t() + (b.double() if bias else 0)
    if relu:
        exp = exp.clamp_min(0)
    out = ops.linear(a.cuda(), w.cuda(), None if b is None else b.cuda(), relu=relu)
    assert out.shape == (m, n) and normwise(out, exp) < 5e-7
    ai = torch.randint(-4, 5, (m, k), generator=g).float()
    wi = torch.randint(-3, 4, (n, k), generator=g).float()
    outi = ops.linear(ai.cuda(), wi.cuda(), None)
    assert torch.equal(outi.cpu(), ai @ wi.t())
    # a strided output view (ldo != n) takes the same kernel
    big = torch.full((m, n + 4), 7.0).cuda()
    ops.linear(a.cuda(), w.cuda(), None if b is None else b.cuda(), relu=relu, out=big[:, :n])
    assert torch.equal(big[:, :n], out) and float(big[:, n:].min()) == 7.0


def test_linear_bf16x3_weight_planes_follow_in_place_updates(rg):
    """The three bf16 planes of a weight are cached per storage / version: an optimizer step (in-place) must invalidate them."""
    _, ops = rg
    torch.manual_seed(0)
    x = torch.randn(4096, 64).cuda()
    w = torch.nn.Parameter(torch.randn(96, 64).cuda())
    y0 = ops.linear(x, w.detach()[:, :], None)
    with torch.no_grad():
        w.mul_(2.0)
    y1 = ops.linear(x, w.detach()[:, :], None)
    assert normwise(y1, 2.0 * y0.double()) < 1e-6
    w.data.mul_(0.5)                                            # .data bypasses the version counter ...
    ops.invalidate_weight_caches()                              # ... so the documented remedy is needed
    y2 = ops.linear(x, w.detach()[:, :], None)
    assert normwise(y2, y0.double()) < 1e-6


def test_linear_split_weights_and_views(rg):
    """One launch, two projections of the same input taken as column views of one weight (how MPNNConv gets P|Q)."""
    _, ops = rg
    g = torch.Generator().manual_seed(3)
    c, d = 64, 144
    x = torch.randn(500, c, generator=g)
    W = torch.randn(d, 2 * c + 16, generator=g)
    b = torch.randn(d, generator=g)
    Wc = W.cuda()
    out = ops.linear(x.cuda(), Wc[:, :c], b.cuda(), w2=Wc[:, c:2 * c])
    exp = torch.cat([x.double() @ W[:, :c].double().t() + b.double(), x.double() @ W[:, c:2 * c].double().t()], 1)
    assert out.shape == (500, 2 * d)
    assert normwise(out, exp) < 2e-6


def test_linear_is_an_exact_fma_chain_on_integers(rg):
    """Integer-valued data: the MFMA result must be exact (and transpose mistakes cannot hide: asymmetric W)."""
    _, ops = rg
    m, k, n = 200, 40, 70
    a = torch.arange(m * k, dtype=torch.float32).reshape(m, k) % 7 - 3
    w = (torch.arange(n * k, dtype=torch.float32).reshape(n, k) % 5 - 2) * (torch.arange(n).reshape(n, 1) % 3 + 1)
    out = ops.linear(a.cuda(), w.cuda(), None)
    assert torch.equal(out.cpu(), a @ w.t())


# ------------------------------------------------------------------------------------ reference known answers
def test_get_mlp_known_answer(rg):
    gnn, _ = rg                                                 # test_gnn.py:9-25
    mlp = gnn.get_mlp(2, 3, [5], False).cuda()
    set_ones(mlp, gnn.Linear)
    mlp.cuda()
    x = torch.tensor([1, 1], dtype=torch.float32).cuda()
    assert mlp[0].weight.shape == (5, 2) and mlp[2].weight.shape == (3, 5)
    assert (mlp(x).detach().cpu().numpy() == np.array([10, 10, 10])).all()


def test_det_net_basic_constructors(rg):
    gnn, _ = rg                                                 # test_gnn.py:28-39
    m = gnn.DetNetBasic(gnn.GNNArchitectureConfig(2, 3, [5], [3], [3], conv_layer_type="MPNNConv"))
    assert isinstance(m.convs[0], gnn.MPNNConv)
    m = gnn.DetNetBasic(gnn.GNNArchitectureConfig(2, 3, [2], [3], [3], conv_layer_type="RadarPointGNNConv"))
    assert isinstance(m.convs[0], gnn.RadarPointGNNConv)
    with pytest.raises(Exception, match="invalid GNN conv layer type"):
        gnn.DetNetBasic(gnn.GNNArchitectureConfig(2, 3, [2], [3], [3], conv_layer_type="GATConv"))


def test_radar_point_gnn_conv_mlps(rg):
    gnn, _ = rg                                                 # test_gnn.py:42-76
    conv = gnn.RadarPointGNNConv(2, 1, "max", 2, 1)
    set_ones(conv.pre_mlp, gnn.Linear); set_ones(conv.post_mlp, gnn.Linear)
    conv.cuda()
    _ = conv.pre_mlp(torch.tensor([[1, 1, 1], [2, 2, 2]], dtype=torch.float32).cuda())
    _ = conv.post_mlp(torch.tensor([[1] * 5, [2] * 5], dtype=torch.float32).cuda())
    assert len(conv.pre_mlp) == 3 and len(conv.post_mlp) == 1


def test_mpnn_conv_mlps_known_answer(rg):
    gnn, _ = rg                                                 # test_gnn.py:79-116
    conv = gnn.MPNNConv(2, 4, 3, post_layers=2)
    set_ones(conv.pre_mlp, gnn.Linear); set_ones(conv.post_mlp, gnn.Linear)
    conv.cuda()
    pre = conv.pre_mlp(torch.tensor([[1.0] * 7, [2.0] * 7]).cuda())
    post = conv.post_mlp(torch.tensor([[1.0] * 9, [2.0] * 9]).cuda())
    assert len(conv.pre_mlp) == 1 and len(conv.post_mlp) == 3
    assert pre[0].tolist() == [7.0] * 7
    assert post[1].tolist() == [72.0] * 4


def test_mpnn_conv_forward_known_answer(rg):
    gnn, _ = rg                                                 # test_gnn.py:119-172 -> 436
    conv = gnn.MPNNConv(2, 4, 3, post_layers=2, aggr="max")
    set_ones(conv.pre_mlp, gnn.Linear); set_ones(conv.post_mlp, gnn.Linear)
    conv.cuda()
    x = torch.tensor([[1, 1], [2, 2]], dtype=torch.float32).cuda()
    ei = torch.tensor([[0, 1, 0], [1, 0, 1]], dtype=torch.long).cuda()
    ea = torch.tensor([[3, 3, 3], [4, 4, 4], [1, 1, 1]], dtype=torch.float32).cuda()
    out = conv.forward(x, ei, ea)
    assert out[1].tolist() == [436.0] * 4


def _hand_cases():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gnn_hand_vectors
    return gnn_hand_vectors.CASES


@pytest.mark.parametrize("case", _hand_cases(), ids=lambda c: c["name"])
def test_hand_derived_asymmetric_vectors_on_device(rg, case):
    """tests/gnn_hand_vectors.py (derivation written out there): asymmetric weights, a duplicate edge, a target without
    incoming edges, max / mean / add, RadarPointGNNConv's residual -- through the HIP modules."""
    gnn, _ = rg
    layer = (gnn.MPNNConv(1, 1, 1, aggr=case["aggr"]) if case["kind"] == "MPNNConv"
             else gnn.RadarPointGNNConv(1, 1, aggr=case["aggr"]))
    layer.load_state_dict({k[2:]: torch.tensor(v, dtype=torch.float32) for k, v in case["state_dict"].items()})
    layer.cuda()
    x = torch.tensor(case["x"], dtype=torch.float32).cuda()
    ei = torch.tensor(case["edge_index"], dtype=torch.int64).cuda()
    ea = torch.tensor(case["edge_attr"], dtype=torch.float32).cuda()
    with torch.no_grad():
        out = layer(x, ei, ea)
    np.testing.assert_allclose(out.cpu().numpy(), np.array(case["expected"]), rtol=2e-6, atol=0)


def test_mpnn_conv_edge_encoder_known_answer(rg):
    gnn, _ = rg                                                 # test_gnn.py:175-221 -> 23
    conv = gnn.MPNNConv(1, 4, 2, use_edge_encoder=True)
    set_ones(conv.pre_mlp, gnn.Linear); set_ones(conv.post_mlp, gnn.Linear)
    conv.edge_encoder.weight = torch.nn.Parameter(torch.ones_like(conv.edge_encoder.weight) * 2)
    conv.edge_encoder.bias = torch.nn.Parameter(torch.zeros_like(conv.edge_encoder.bias))
    conv.cuda()
    x = torch.tensor([[1], [2]], dtype=torch.float32).cuda()
    ei = torch.tensor([[0, 1], [1, 0]], dtype=torch.long).cuda()
    ea = torch.tensor([[1, 1], [2, 2]], dtype=torch.float32).cuda()
    out = conv.forward(x, ei, ea)
    assert conv.edge_encoder(ea)[0].item() == 4
    assert conv.pre_mlp[0].weight[0].shape[0] == 3
    assert out[1, 0].item() == 23


# ------------------------------------------------------------------------------------ layers vs oracle
def random_graph(n, e, seed, isolated=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    if isolated:
        ei[1][ei[1] >= n - 5] = 0                               # nodes n-5..n-1 receive nothing -> must aggregate to 0
    ei[:, 1] = ei[:, 0]                                         # a duplicate edge
    return ei


@pytest.mark.parametrize("aggr", ["max", "mean", "add"])
@pytest.mark.parametrize("cin,cout,de,pre,post,enc", [(16, 24, 4, 1, 1, False), (224, 224, 16, 1, 1, False),
                                                      (10, 7, 3, 1, 2, False), (12, 20, 5, 2, 1, False),
                                                      (8, 8, 6, 1, 1, True), (128, 64, 16, 3, 2, True),
                                                      (64, 32, 24, 1, 1, False), (200, 64, 10, 1, 1, False)])
def test_mpnn_conv_vs_oracle(rg, aggr, cin, cout, de, pre, post, enc):
    gnn, _ = rg
    torch.manual_seed(cin + cout)
    conv = gnn.MPNNConv(cin, cout, de, aggr=aggr, pre_layers=pre, post_layers=post, use_edge_encoder=enc)
    n, e = 400, 3000
    ei = random_graph(n, e, 1)
    x = torch.randn(n, cin)
    ea = torch.randn(e, de)
    sd = {"c." + k: v for k, v in cpu_sd(conv).items()}
    exp = G.mpnn_conv(x.double(), ei, ea.double(), {k: v.double() for k, v in sd.items()}, "c.", aggr)
    got = conv.cuda()(x.cuda(), ei.cuda(), ea.cuda())
    assert normwise(got, exp) < RTOL
    m_exp_isolated = exp[n - 5:]
    assert torch.isfinite(got).all() and m_exp_isolated.shape[0] == 5


@pytest.mark.parametrize("aggr", ["max", "mean", "add"])
@pytest.mark.parametrize("c,de,pre,post", [(16, 4, 1, 1), (224, 16, 1, 1), (9, 3, 2, 2)])
def test_radar_point_gnn_conv_vs_oracle(rg, aggr, c, de, pre, post):
    gnn, _ = rg
    torch.manual_seed(c)
    conv = gnn.RadarPointGNNConv(c, de, aggr=aggr, pre_layers=pre, post_layers=post)
    n, e = 300, 2500
    ei = random_graph(n, e, 2)
    x = torch.randn(n, c)
    ea = torch.randn(e, de)
    sd = {"c." + k: v.double() for k, v in cpu_sd(conv).items()}
    exp = G.radar_point_gnn_conv(x.double(), ei, ea.double(), sd, "c.", aggr)
    got = conv.cuda()(x.cuda(), ei.cuda(), ea.cuda())
    assert normwise(got, exp) < RTOL


def test_empty_segments_are_exactly_zero(rg):
    _, ops = rg
    n, d = 50, 16
    ei = torch.tensor([[1, 2, 3], [0, 0, 4]], dtype=torch.long).cuda()
    rowptr, src, perm = ops.csr_by_target(ei, n)
    Q = torch.randn(n, d).cuda()
    P = torch.randn(n, d).cuda()
    for aggr in ("max", "mean", "add"):
        m = ops.mpnn_aggregate(P, None, Q, None, None, rowptr, src, aggr)
        assert (m[1:4] == 0).all() and (m[5:] == 0).all()
        assert (m[0] != 0).any()


def test_batchnorm_module_matches_torch(rg):
    gnn, _ = rg
    torch.manual_seed(0)
    x = torch.randn(1000, 37) * 3 + 1.5
    bn = gnn.BatchNorm(37)
    ref = torch.nn.BatchNorm1d(37)
    with torch.no_grad():
        bn.module.weight.uniform_(0.5, 1.5); bn.module.bias.uniform_(-1, 1)
        ref.weight.copy_(bn.module.weight); ref.bias.copy_(bn.module.bias)
    bn.cuda()
    y = bn(x.cuda())
    y_ref = ref(x)
    assert normwise(y, y_ref) < 1e-5
    assert normwise(bn.module.running_mean, ref.running_mean) < 1e-5
    assert normwise(bn.module.running_var, ref.running_var) < 1e-5
    assert bn.module.num_batches_tracked.item() == 1
    bn.eval(); ref.eval()
    assert normwise(bn(x.cuda()), ref(x)) < 1e-5
    assert bn.module.num_batches_tracked.item() == 1


# ------------------------------------------------------------------------------------ whole model
def shipped_config(gnn, n_conv=5, k_classes=6, conv_type="MPNNConv", bn_in_mlps=False, aggr="max"):
    dims = [224, 224, 128, 64, 32][:n_conv] if conv_type == "MPNNConv" else [224] * n_conv
    return gnn.GNNArchitectureConfig(5, 2, dims, [k_classes], [16, 5], True, True, [32, 64, 128, 224], [4, 8, 16],
                                     conv_type, bn_in_mlps, 1, 1, False, aggr)


def frame_graph(routine="knn", k=20, r=1.0, idx=0):
    f = synthetic.radarscenes_frame(idx)
    g = go.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, routine, k, r,
                             ["rcs", "velocity_vector", "time_index", "degree"], ["relative_position"], "directed")
    return torch.from_numpy(g["x"]), torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_attr"])


@pytest.mark.parametrize("conv_type,routine,bn_in_mlps,aggr", [("MPNNConv", "knn", False, "max"),
                                                               ("MPNNConv", "radius", False, "max"),
                                                               ("MPNNConv", "knn", True, "mean"),
                                                               ("RadarPointGNNConv", "knn", False, "max"),
                                                               ("MPNNConv", "radius", True, "add")])
def test_det_net_basic_vs_oracle(rg, conv_type, routine, bn_in_mlps, aggr):
    """Shipped RadarScenes architecture (configurations/configuration_radarscenes.yml:28-41) on a 3000-point frame,
    module in training mode like the reference (batch statistics in every BatchNorm)."""
    gnn, _ = rg
    torch.manual_seed(0)
    model = gnn.DetNetBasic(shipped_config(gnn, conv_type=conv_type, bn_in_mlps=bn_in_mlps, aggr=aggr))
    x, ei, ea = frame_graph(routine)
    sd = cpu_sd(model)
    c64, b64, hidden = G.det_net_basic(x, ei, ea, sd, conv_type, aggr, training=True, dtype=torch.float64, return_hidden=True)
    c32, b32 = G.det_net_basic(x, ei, ea, sd, conv_type, aggr, training=True, dtype=torch.float32)
    model.cuda()
    c, bb = model(x.cuda(), ei.cuda(), ea.cuda())
    e_c, e_b = normwise(c, c64), normwise(bb, b64)
    o_c, o_b = normwise(c32, c64), normwise(b32, b64)
    print(f"\n[{conv_type} {routine} bn_mlps={bn_in_mlps} {aggr}] HIP vs f64: cls {e_c:.2e} box {e_b:.2e} | "
          f"torch-f32 oracle vs f64: cls {o_c:.2e} box {o_b:.2e}")
    assert e_c < RTOL and e_b < RTOL
    # running statistics moved exactly like torch's BatchNorm1d would move them
    rm, rv = G.bn_running_update(hidden[0], sd["batch_norms.0.module.running_mean"].double(),
                                 sd["batch_norms.0.module.running_var"].double())
    assert normwise(model.batch_norms[0].module.running_mean, rm) < 1e-5
    assert normwise(model.batch_norms[0].module.running_var, rv) < 1e-5
    assert model.batch_norms[0].module.num_batches_tracked.item() == 1


def test_det_net_basic_eval_mode_and_state_dict_roundtrip(rg):
    gnn, _ = rg
    torch.manual_seed(1)
    model = gnn.DetNetBasic(shipped_config(gnn))
    with torch.no_grad():
        for bn in model.batch_norms:
            bn.module.running_mean.normal_(0, 0.1); bn.module.running_var.uniform_(0.5, 2.0)
    sd = cpu_sd(model)
    clone = gnn.DetNetBasic(shipped_config(gnn))
    clone.load_state_dict(sd)                                  # reference-keyed checkpoint loads
    x, ei, ea = frame_graph("knn", k=10)
    exp_c, exp_b = G.det_net_basic(x, ei, ea, sd, training=False, dtype=torch.float64)
    clone.cuda().eval()
    c, bb = clone(x.cuda(), ei.cuda(), ea.cuda())
    assert normwise(c, exp_c) < RTOL and normwise(bb, exp_b) < RTOL
    assert clone.batch_norms[0].module.num_batches_tracked.item() == 0


def test_batched_frames_forward(rg):
    """PyG-style batch of 4 frames (utils/data_handling.py:30): BatchNorm statistics span the whole batch."""
    gnn, _ = rg
    torch.manual_seed(2)
    model = gnn.DetNetBasic(shipped_config(gnn, n_conv=4))
    graphs = []
    for i in range(4):
        f = synthetic.nuscenes_frame(i)
        graphs.append(go.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, "radius", None, 6.0,
                                           ["rcs", "velocity_vector", "time_index", "degree"], ["relative_position"],
                                           "directed"))
    b = go.collate(graphs)
    x, ei, ea = torch.from_numpy(b["x"]), torch.from_numpy(b["edge_index"]), torch.from_numpy(b["edge_attr"])
    exp_c, exp_b = G.det_net_basic(x, ei, ea, cpu_sd(model), dtype=torch.float64)
    model.cuda()
    c, bb = model(x.cuda(), ei.cuda(), ea.cuda())
    assert normwise(c, exp_c) < RTOL and normwise(bb, exp_b) < RTOL
    p = torch.softmax(c, 1)
    from radargnn_amd import ops
    assert normwise(ops.softmax_rows(c), p) < 1e-6


def test_forward_rejects_cpu_tensors(rg):
    gnn, _ = rg
    conv = gnn.MPNNConv(2, 4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        conv(torch.zeros(2, 2), torch.zeros(2, 1, dtype=torch.long), torch.zeros(1, 3))


def test_few_row_dense_layers_narrow_tiles_and_parallel_split_k(rg, monkeypatch):
    """One frame's worth of rows on the LDS-DMA kernel: the column tile narrows so that more work-groups share the layer (same
    accumulation order: bit-identical to the wide tile), and the k-loop is cut over parallel work-groups whose partial
    accumulators are added in a fixed order (deterministic, last-bit different, same error class against float64)."""
    _, ops = rg
    g = torch.Generator().manual_seed(17)
    m, k1, k2, n = 3000, 224, 464, 224
    a1, a2 = torch.randn(m, k1, generator=g).cuda(), torch.randn(m, k2, generator=g).cuda()
    w, b = (torch.randn(n, k1 + k2, generator=g) * 0.05).cuda(), torch.randn(n, generator=g).cuda()
    rows = torch.randperm(m, generator=g)[:1700].sort().values.int().cuda()
    lst = torch.zeros(m, dtype=torch.int32, device="cuda"); lst[:1700] = rows
    cnt = torch.tensor([1700], dtype=torch.int64, device="cuda")
    exp = torch.cat([a1, a2], 1).double() @ w.double().t() + b.double()

    def run():
        dense, st = ops.linear(a1, w, b, a2=a2, relu=False, want_stats=True)
        sub = torch.zeros(m, n, device="cuda")
        ops.linear(a1, w, b, a2=a2, out=sub, row_index=lst, m_dev=cnt)
        return dense, st, sub

    d0, s0, u0 = run()
    assert torch.equal(d0, run()[0]) and torch.equal(u0, run()[2])               # deterministic
    assert normwise(d0, exp) < 1e-6 and normwise(u0[rows.long()], exp[rows.long().cpu()]) < 1e-6
    monkeypatch.setenv("RGNN_DMA_NOPSK", "1")
    __import__("radargnn_amd.ops").ops.reload_env()
    d1, s1, u1 = run()                                                           # narrow tiles, undivided k-loop
    monkeypatch.setenv("RGNN_DMA_NO_SMALL_M", "1")
    __import__("radargnn_amd.ops").ops.reload_env()
    d2, s2, u2 = run()                                                           # the tile width a full batch would get
    assert torch.equal(d1, d2) and torch.equal(u1, u2) and torch.equal(s1, s2)
    assert normwise(d0, d1.double()) < 1e-6 and not torch.equal(d0, d1)          # split-K: other rounding, same class


@pytest.mark.parametrize("m,sub,k1,k2,n,relu_in", [(6000, 0, 224, 0, 464, True), (5000, 2777, 224, 464, 224, True),
                                                    (4096, 4000, 128, 272, 128, False), (300, 0, 64, 0, 96, True),
                                                    (9000, 5000, 32, 32, 272, True), (3000, 0, 48, 0, 40, True),
                                                    (2000, 0, 36, 0, 68, True), (100, 0, 64, 0, 96, True)])
def test_linear_applies_the_previous_batchnorm_inside_its_a_operand(rg, monkeypatch, m, sub, k1, k2, n, relu_in):
    """``a1_affine``: the layer reads act((a1 - mean) * g + t) -- the BatchNorm + ReLU of the layer before -- without that
    tensor ever being written.  The LDS-DMA kernel applies it to its A fragments (same subtract, fmaf and max as the stand-alone
    pass), so the fused launch and  scale_shift_act -> linear  give the same bits; launches the kernel cannot take (odd
    widths, few rows) fall back to exactly that pair."""
    _, ops = rg
    g = torch.Generator().manual_seed(m + k1)
    a1 = torch.randn(m, k1, generator=g).cuda()
    a2 = torch.randn(m, k2, generator=g).cuda() if k2 else None
    w = (torch.randn(n, k1 + k2, generator=g) / np.sqrt(k1 + k2)).cuda()
    b = torch.randn(n, generator=g).cuda()
    aff = torch.stack([torch.randn(k1, generator=g), torch.rand(k1, generator=g) + 0.5, torch.randn(k1, generator=g)]).cuda()   # mean_hi, g, t
    kw = {}
    if sub:
        rows = torch.randperm(m, generator=g)[:sub].sort().values
        lst = torch.full((m,), -7, dtype=torch.int32)
        lst[:sub] = rows.to(torch.int32)
        kw = dict(row_index=lst.cuda(), m_dev=torch.tensor([sub]).cuda())
    else:
        rows = torch.arange(m)
    out, st = {}, {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSE_A1_AFFINE", fused)
        before = ops.COUNTERS["fused_a1_affine"]
        o = torch.full((m, n), 777.0).cuda()
        s = torch.zeros((max(ops.stat_panels(m), 1), ops.STAT_ROWS, n)).cuda()
        ops.linear(a1, w, b, a2=a2, relu=True, out=o, stats_out=s, a1_affine=aff, a1_relu=relu_in, **kw)
        took = ops.COUNTERS["fused_a1_affine"] - before
        planes = m >= ops.BF16X3_MIN_ROWS and n > ops.BF16X3_MIN_COLS and n % 4 == 0      # bf16x3 kernels (weight planes given)
        eligible = (k1 % 16 == 0 and k2 % 16 == 0) if planes else (k1 % 4 == 0 and k2 % 4 == 0)   # LDS-DMA / fp32 kernel
        assert took == (1 if fused and eligible else 0)
        out[fused], st[fused] = o.cpu(), s.cpu()
    assert torch.equal(out[True], out[False])
    assert torch.equal(st[True], st[False])
    h = ops.apply_table_reference(a1.cpu(), aff.cpu())
    if relu_in:
        h = h.clamp_min(0)
    a = h if a2 is None else torch.cat([h, a2.double().cpu()], 1)
    exp = (a[rows] @ w.double().cpu().t() + b.double().cpu()).clamp_min(0)
    assert normwise(out[True][rows], exp) < 2e-6
    mask = torch.ones(m, dtype=torch.bool)
    mask[rows] = False
    assert torch.all(out[True][mask] == 777.0)


def test_batchnorm_finalize_over_two_row_subset_launches_reads_live_panels_only(rg):
    """rgnn_batchnorm_finalize_parts: the statistics of one layer output come from two row-subset launches with a buffer each;
    only the ceil(rows / 128) panels a launch wrote are read (everything else is poisoned with NaN here), and the result equals
    the one-buffer finalize over zero-filled buffers (up to the order the panels are added in)."""
    _, ops = rg
    g = torch.Generator().manual_seed(9)
    m, n, c = 5000, 224, 224
    x = torch.randn(m, c, generator=g).cuda()
    w = (torch.randn(n, c, generator=g) / 15).cuda()
    b = torch.randn(n, generator=g).cuda()
    perm = torch.randperm(m, generator=g)
    na = 3217
    panels = ops.stat_panels(m)
    rows_a = torch.full((m,), -3, dtype=torch.int32); rows_a[:na] = perm[:na].sort().values.to(torch.int32)
    rows_b = torch.full((m,), -3, dtype=torch.int32); rows_b[:m - na] = perm[na:].sort().values.to(torch.int32)
    cnt_a, cnt_b = torch.tensor([na]).cuda(), torch.tensor([m - na]).cuda()
    gamma, beta = torch.rand(n, generator=g).cuda() + 0.5, torch.randn(n, generator=g).cuda()
    res = {}
    for poison in (True, False):
        buf = torch.full((2 * panels, ops.STAT_ROWS, n), float("nan") if poison else 0.0).cuda()
        out = torch.empty(m, n).cuda()
        ops.linear(x, w, b, out=out, row_index=rows_a.cuda(), m_dev=cnt_a, stats_out=buf[:panels])
        ops.linear(x, w, b, out=out, row_index=rows_b.cuda(), m_dev=cnt_b, stats_out=buf[panels:])
        rm, rv, nb = torch.zeros(n).cuda(), torch.ones(n).cuda(), torch.zeros((), dtype=torch.int64).cuda()
        st = ops.StatParts([(buf[:panels], cnt_a), (buf[panels:], cnt_b)]) if poison else buf
        res[poison] = (ops.batchnorm_finalize(st, m, n, gamma, beta, rm, rv, nb, True, 0.1, 1e-5), rm, rv, int(nb))
    for a_, b_ in zip(res[True], res[False]):        # (the panels meet in another order: last-bit differences, nothing more)
        if torch.is_tensor(a_):
            assert bool(torch.isfinite(a_).all())
            np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), rtol=2e-6, atol=1e-6)
        else:
            assert a_ == b_
    ref = torch.nn.functional.batch_norm(out.double(), None, None, gamma.double(), beta.double(), True, 0.1, 1e-5)
    got = ops.apply_table_reference(out, res[True][0])
    assert normwise(got, ref) < 1e-6
    # one part, no live count: all panels
    one = ops.batchnorm_finalize(ops.StatParts([(buf, None)]), m, n, gamma, beta, None, None, None, True, 0.1, 1e-5)
    np.testing.assert_allclose(one.cpu().numpy(), res[False][0].cpu().numpy(), rtol=2e-6, atol=1e-6)


def test_model_with_fused_batchnorm_apply_equals_the_layer_by_layer_model(rg, monkeypatch):
    """DetNetBasic hands every MPNN layer the scale / shift of the BatchNorm before it instead of the normalised features
    (train mode: batch statistics; eval mode: running statistics): same outputs, bit for bit, as with
    RGNN_NO_FUSED_BN_APPLY=1, and the fused launches really ran."""
    gnn, ops = rg
    torch.manual_seed(3)
    model = gnn.DetNetBasic(shipped_config(gnn, n_conv=4)).cuda()
    x, ei, ea = (t.cuda() for t in frame_graph("radius", r=4.0))
    for training in (True, False):
        model.train(training)
        res = {}
        for fused in (True, False):
            monkeypatch.setattr(ops, "FUSE_A1_AFFINE", fused)
            before = ops.COUNTERS["fused_a1_affine"]
            with torch.no_grad():
                res[fused] = [t.clone() for t in model(x, ei, ea)]
            took = ops.COUNTERS["fused_a1_affine"] - before
            assert (took >= 6) if fused else (took == 0)       # layers 2..4, two or three dense launches each
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def test_both_heads_in_one_launch_equal_the_separate_heads(rg, monkeypatch):
    """Inference: the first Linears of the classification and the regression head run as ONE launch behind the last
    BatchNorm (its scale / shift applied inside the kernel, ReLU on the regression head's columns only): same logits and
    boxes, bit for bit, as normalise -> head -> head; also with the ReLU on the other head, on both and on neither."""
    gnn, ops = rg
    import radargnn_amd.gnn.gnn_models as GM
    x, ei, ea = (t.cuda() for t in frame_graph("radius", r=4.0))
    for cls_dims, reg_dims in (([6], [16, 5]), ([12, 6], [5]), ([8, 6], [16, 5]), ([6], [5])):
        torch.manual_seed(4)
        cfg = shipped_config(gnn, n_conv=4)
        cfg.classification_head_layer_dimensions, cfg.regression_head_layer_dimensions = cls_dims, reg_dims
        model = gnn.DetNetBasic(cfg).cuda()
        res = {}
        for fused in (True, False):
            monkeypatch.setattr(GM, "FUSE_HEADS", fused)
            with torch.no_grad():
                res[fused] = [t.clone() for t in model(x, ei, ea)]
        assert res[True][0].shape == res[False][0].shape and res[True][1].shape == res[False][1].shape
        assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
        # the fused weights follow an in-place update of a head
        with torch.no_grad():
            model.regression_head[0].weight.mul_(1.5)
            monkeypatch.setattr(GM, "FUSE_HEADS", True)
            a = [t.clone() for t in model(x, ei, ea)]
            monkeypatch.setattr(GM, "FUSE_HEADS", False)
            b_ = [t.clone() for t in model(x, ei, ea)]
        assert torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1])


def test_source_term_only_on_rows_with_outgoing_edges(rg, monkeypatch):
    """A directed graph in which some nodes have no outgoing edges: the source-term GEMM skips their rows
    (TargetCSR.source_rows from the out-degrees); outputs equal those of the all-rows launch bit for bit (with the k-loop
    undivided: at this size the parallel split-K factor follows the row count of a launch)."""
    monkeypatch.setenv("RGNN_DMA_NOPSK", "1")
    __import__("radargnn_amd.ops").ops.reload_env()
    gnn, ops = rg
    from radargnn_amd.gnn.mpnn_layers import TargetCSR
    torch.manual_seed(9)
    model = gnn.DetNetBasic(shipped_config(gnn, n_conv=2, aggr="max")).cuda().eval()
    n, e = 2500, 15000
    gen = torch.Generator().manual_seed(2)
    src = torch.randint(0, n // 2, (e,), generator=gen)                          # only the first half of the nodes are sources
    dst = torch.randint(0, n, (e,), generator=gen)
    keep = src != dst
    ei = torch.unique(torch.stack([src[keep], dst[keep]]), dim=1).cuda()
    x, ea = torch.randn(n, 5, generator=gen).cuda(), torch.randn(ei.shape[1], 2, generator=gen).cuda()
    g_all = TargetCSR(ei, n, all_sources=True)                                   # (a hint that happens to be false: dense launch)
    g_src = TargetCSR(ei, n)
    lst, cnt = g_src.source_rows()
    assert int(cnt.item()) == int(torch.unique(ei[0]).numel())
    assert torch.equal(torch.sort(lst[:int(cnt.item())].long()).values, torch.unique(ei[0]))
    c0, b0 = model.forward_graph(x, g_all, g_all.sort_edge_attr(ea))
    c1, b1 = model.forward_graph(x, g_src, g_src.sort_edge_attr(ea))
    assert torch.equal(c0, c1) and torch.equal(b0, b1)


@pytest.mark.parametrize("aggr,pre", [("max", 1), ("mean", 1), ("add", 2)])
def test_visiting_order_never_changes_results(rg, aggr, pre):
    """The grid-cell visiting order (CSR laid out by target rank) is a scheduling choice: bit-identical outputs."""
    gnn, ops = rg
    from radargnn_amd.gnn.mpnn_layers import TargetCSR
    torch.manual_seed(5)
    cfg = shipped_config(gnn, n_conv=3, aggr=aggr)
    cfg.conv_pre_mlp_layer_number = pre
    model = gnn.DetNetBasic(cfg).cuda().eval()
    x, ei, ea = frame_graph("radius")
    x, ei, ea = x.cuda(), ei.cuda(), ea.cuda()
    c0, b0 = model(x, ei, ea)
    order = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(1)).to(torch.int32).cuda()
    g = TargetCSR(ei, x.shape[0], order=order)
    c1, b1 = model.forward_graph(x, g, g.sort_edge_attr(ea))
    assert torch.equal(c0, c1) and torch.equal(b0, b1)
    # the ordered CSR really is laid out by rank
    rank = ops.invert_permutation(order).cpu().numpy()
    key = rank[ei[1].cpu().numpy()].astype(np.int64) * (1 << 32) + np.arange(ei.shape[1])
    assert np.array_equal(g.perm.cpu().numpy(), np.argsort(key, kind="stable"))


def test_batchnorm_on_zero_and_one_rows_behaves_like_torch():
    """torch's train-mode batch_norm: one row is an error, zero rows pass through with the running statistics untouched and
    num_batches_tracked counting the call -- an edge MLP with BatchNorm on a graph without edges (reference: gnn_models.py:137-178
    with batch_norm_in_mlps).  The same through run_mlp's fused Linear + BatchNorm + ReLU."""
    import torch.nn as tnn
    from radargnn_amd import gnn
    from radargnn_amd.gnn.linear import BatchNorm, Linear, run_mlp
    bn = BatchNorm(4).cuda().train()
    ref = tnn.BatchNorm1d(4).train()
    with torch.no_grad():
        y = bn(torch.empty(0, 4, device="cuda")); ref(torch.empty(0, 4))
    assert y.shape == (0, 4)
    assert int(bn.module.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    assert torch.equal(bn.module.running_mean.cpu(), ref.running_mean) and torch.equal(bn.module.running_var.cpu(), ref.running_var)
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        bn(torch.randn(1, 4, device="cuda"))
    seq = tnn.Sequential(Linear(2, 4), BatchNorm(4), tnn.ReLU(), Linear(4, 8)).cuda().train()
    with torch.no_grad():
        out, _ = run_mlp(seq, torch.empty(0, 2, device="cuda"))
    assert out.shape == (0, 8) and int(seq[1].module.num_batches_tracked) == 1


@pytest.mark.parametrize("m,k0,relu3", [(1, 5, True), (255, 5, True), (3000, 4, False), (70001, 8, True)])
def test_three_layer_embedding_in_one_pass(m, k0, relu3):
    """rgnn_embed3 (x -> 32 -> 64 -> 128 with ReLUs, the shipped node embedding) against float64 torch: 2e-6 norm-wise, like the
    dense kernels it replaces; its bound covers max |out|."""
    from radargnn_amd import ops
    g = torch.Generator().manual_seed(m)
    x = (torch.randn(m, k0, generator=g) * torch.tensor([30.0, 5, 5, 1, 8, 1, 1, 1])[:k0]).cuda()
    ws = [(torch.randn(32, k0, generator=g) / k0 ** 0.5).cuda(), (torch.randn(64, 32, generator=g) / 32 ** 0.5).cuda(),
          (torch.randn(128, 64, generator=g) / 8).cuda()]
    bs = [torch.randn(n, generator=g).cuda() for n in (32, 64, 128)]
    with ops.bound_tracking(x.device):
        out = ops.embed3(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], relu3)
        bound = float(ops.bound_of(out).max())
    h = x.double()
    for i, (w, b) in enumerate(zip(ws, bs)):
        h = h @ w.double().t() + b.double()
        if i < 2 or relu3:
            h = torch.relu(h)
    err = ((out.double() - h).abs().max() / h.abs().max()).item()
    assert err <= 2e-6, err
    assert bound >= float(out.abs().max()) and bound <= 1.001 * float(out.abs().max()) + 1e-30
    assert ops.embed3(x, torch.randn(16, k0, device="cuda"), None, torch.randn(64, 16, device="cuda"), None, ws[2], None, True) is None
