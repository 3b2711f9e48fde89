"""Train-mode BatchNorm on columns whose spread is tiny against their mean.

r03 applied BatchNorm as ``x * scale + shift`` with ``shift = beta - mean * scale`` rounded to float and took the variance
from float32 sums of x and x^2 per 128-row panel: on a constant column (all nodes with the same degree in front of an MLP
with BatchNorm inside) the rounding of ``mean * scale`` came out as 1e-4 against the float64 oracle.  (ATen's own float32 CPU
kernel has the same form and the same deviation: 1.2e-4 ... 1.6e-2 measured in this image.)  Now the statistics are kept as
{count, mean, M2} about a pivot taken from the data and the table applies ``(x - mean_hi) g + t``: a constant column comes out as
beta EXACTLY, and a near-constant one to float32 accuracy of the normalised value."""
import copy

import numpy as np
import pytest
import torch

from oracle import gnn_oracle as G
from oracle import graph_oracle as go
from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu


def normwise(a, b) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def _bn64(x: torch.Tensor, gamma, beta, eps=1e-5):
    x = x.double().cpu()
    mu, var = x.mean(0), x.var(0, unbiased=False)
    return (x - mu) / torch.sqrt(var + eps) * gamma.double().cpu() + beta.double().cpu()


def _columns(m: int, seed: int) -> torch.Tensor:
    """[m, 8] float32: constant columns (values with full 24-bit significands), near-constant ones (|mean| / std from 4e2 to
    9e4, values exact in float32), and ordinary ones."""
    g = torch.Generator().manual_seed(seed)
    x = torch.empty(m, 8)
    x[:, 0] = 17.123456
    x[:, 1] = -3.1415927e3
    x[:, 2] = 20.0
    x[:7, 2] = 21.0                                 # degrees: all 20 but seven
    x[:, 3] = 200.0
    x[m // 2, 3] = 201.0                            # one row differs: |mean| / std = 200 sqrt(m)
    x[:, 4] = 1.0e4 + torch.randint(0, 3, (m,), generator=g).float() * 2.0 ** -10     # three neighbouring floats around 1e4
    x[:, 5] = torch.randn(m, generator=g)
    x[:, 6] = torch.randn(m, generator=g) * 30.0 + 500.0
    x[:, 7] = 0.0                                    # a dead unit
    return x


@pytest.mark.parametrize("m", [97, 3000, 192000])
def test_stand_alone_batchnorm_on_constant_and_near_constant_columns(m):
    """rgnn_column_stats -> rgnn_batchnorm_finalize -> rgnn_scale_shift_act against float64 on the same float32 input."""
    from radargnn_amd import ops
    x = _columns(m, m).cuda()
    n = x.shape[1]
    g = torch.Generator().manual_seed(1)
    gamma, beta = (torch.rand(n, generator=g) + 0.5).cuda(), torch.randn(n, generator=g).cuda()
    rm, rv, nb = torch.zeros(n).cuda(), torch.ones(n).cuda(), torch.zeros((), dtype=torch.int64).cuda()
    table = ops.batchnorm_finalize(ops.column_stats(x), m, n, gamma, beta, rm, rv, nb, True, 0.1, 1e-5)
    y = ops.scale_shift_act(x, table, relu=False)
    ref = _bn64(x, gamma, beta)
    # constant columns: exactly beta
    for c in (0, 1, 7):
        assert torch.equal(y[:, c].cpu(), beta[c].cpu().expand(m)), c
    err = ((y.double().cpu() - ref).abs().max(0).values / ref.abs().max(0).values.clamp_min(1.0))
    assert float(err.max()) < 1e-6, err
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * x.double().mean(0).cpu().numpy(), rtol=1e-6, atol=1e-30)
    np.testing.assert_allclose(rv.cpu().numpy(), 0.9 + 0.1 * x.double().var(0).cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("m,sub", [(5000, 0), (5000, 3100), (200, 0)])
def test_dense_epilogue_statistics_on_constant_and_near_constant_columns(m, sub):
    """The same columns produced by dense launches (small-integer operands: every output element is exact in float32, so the
    float64 reference sees the same matrix), plain and on a row subset, on the LDS-DMA kernel and the fp32 one."""
    from radargnn_amd import ops
    k, n = 32, 96
    g = torch.Generator().manual_seed(m + sub)
    a = torch.randint(-2, 3, (m, k), generator=g).float()
    a[:, 0] = 1.0                                                  # carries the "bias" of the constant columns exactly
    w = torch.zeros(n, k)
    w[:, 1:] = torch.randint(-3, 4, (n, k - 1), generator=g).float()
    w[:32, 1:] = 0.0                                               # columns 0 .. 31: constant = w[:, 0]
    w[:32, 0] = torch.arange(32).float() * 37.0 + 1001.0
    w[32:64, 1:3] = 0.0                                            # columns 32 .. 63: integer-valued around a large offset
    w[32:64, 0] = 4096.0
    b = torch.zeros(n)
    kw, rows = {}, torch.arange(m)
    if sub:
        rows = torch.randperm(m, generator=g)[:sub].sort().values
        lst = torch.full((m,), -5, dtype=torch.int32)
        lst[:sub] = rows.to(torch.int32)
        kw = dict(row_index=lst.cuda(), m_dev=torch.tensor([sub]).cuda())
    gamma, beta = (torch.rand(n, generator=g) + 0.5).cuda(), torch.randn(n, generator=g).cuda()
    for x3 in (True, False):
        ops.USE_BF16X3 = x3
        try:
            out = torch.zeros(m, n).cuda()
            st = torch.zeros((max(ops.stat_panels(m), 1), ops.STAT_ROWS, n)).cuda()
            ops.linear(a.cuda(), w.cuda(), b.cuda(), out=out, stats_out=st, **kw)
        finally:
            ops.USE_BF16X3 = True
        exp = a[rows].double() @ w.double().t()
        assert torch.equal(out[rows.cuda()].double().cpu(), exp)     # (exact: integers below 2^24)
        table = ops.batchnorm_finalize(st, len(rows), n, gamma, beta, None, None, None, True, 0.1, 1e-5)
        y = ops.scale_shift_act(out[rows.cuda()].contiguous(), table, relu=False)
        ref = _bn64(exp, gamma, beta)
        assert torch.equal(y[:, :32].cpu(), beta[:32].cpu().expand(len(rows), 32))
        assert normwise(y, ref) < 1e-6


def _uniform_degree_frame(idx: int, clusters: int = 120, per: int = 8):
    """A frame whose radius graph (r = 1) gives EVERY node the same degree: tight clusters of `per` points on a 10 m grid."""
    rng = np.random.default_rng(900 + idx)
    f = synthetic.radarscenes_frame(idx, n_clusters=1, n_clutter=clusters * per - 35)
    n = f.X.shape[0]
    centres = np.stack([(np.arange(clusters) % 12) * 10.0, (np.arange(clusters) // 12) * 10.0 - 40.0], 1)
    f.X[:] = (np.repeat(centres, per, 0)[:n] + rng.uniform(-0.3, 0.3, (n, 2))).astype(np.float32).astype(np.float64)
    return f


@pytest.mark.parametrize("bn_scope", ["batch", "frame"])
def test_model_whose_only_node_feature_is_a_constant_degree(bn_scope):
    """DetNetBasic with BatchNorm inside its MLPs on graphs where all degrees are equal -- the case the r03 fuzz runs kept
    rediscovering at 1e-4: now within 1e-5 of the float64 oracle, batch-wide and per frame, eager and replayed."""
    from radargnn_amd import frames as fr, gnn
    frames = [_uniform_degree_frame(i) for i in range(4)]
    cfg = fr.GraphSettings(algorithm="radius", r=1.0, node_features=("degree",))
    mcfg = gnn.GNNArchitectureConfig(1, 2, [64, 32], [6], [16, 5], True, True, [32, 64], [4, 8, 16], "MPNNConv", True)
    torch.manual_seed(5)
    model = gnn.DetNetBasic(mcfg)
    with torch.no_grad():
        for name, p in model.named_parameters():                   # (beta != 0, gamma != 1: the shift term takes part)
            if ".module." in name:
                p.copy_(torch.rand_like(p) + 0.5)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().train()
    batch = fr.FrameBatch.from_frames(frames)
    cls, bb, g = fr.HotPath(copy.deepcopy(model), cfg, bn_scope=bn_scope)(batch)
    g.check()
    deg = g.x[:, 0]
    assert bool((deg == deg[0]).all()) and float(deg[0]) == 7.0
    refs = [go.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, "radius", None, 1.0, ["degree"], list(cfg.edge_features), "directed")
            for f in frames]
    if bn_scope == "batch":
        ref = go.collate(refs)
        c64, b64 = G.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]), torch.from_numpy(ref["edge_attr"]),
                                   sd, dtype=torch.float64)
        # the oracle numbers edges per source row; HotPath's graph too (canonical order)
        assert normwise(cls, c64) < 1e-5 and normwise(bb, b64) < 1e-5, (normwise(cls, c64), normwise(bb, b64))
    else:
        ptr = batch.frame_ptr.cpu().numpy()
        for f, r in enumerate(refs):
            c64, b64 = G.det_net_basic(torch.from_numpy(r["x"]), torch.from_numpy(r["edge_index"]), torch.from_numpy(r["edge_attr"]),
                                       sd, dtype=torch.float64)
            sl = slice(int(ptr[f]), int(ptr[f + 1]))
            assert normwise(cls[sl], c64) < 1e-5 and normwise(bb[sl], b64) < 1e-5, (f, normwise(cls[sl], c64), normwise(bb[sl], b64))
    hot = fr.HotPath(copy.deepcopy(model), cfg, bn_scope=bn_scope, use_hip_graphs=True)
    for _ in range(4):
        c3, b3, _ = hot(batch)
    torch.cuda.synchronize()
    assert hot._graph is not None and torch.equal(c3, cls) and torch.equal(b3, bb)


@pytest.mark.parametrize("m", [604, 3000])
def test_batch_norm_with_extreme_rows_at_the_head_of_the_batch(m):
    """A handful of rows 20 - 50 spreads away from the rest AT THE HEAD of the batch (a tiny frame of very different points in front:
    tools/fuzz_hot_path.py seed 202 case 37).  The statistics are sums about a pivot, and r05 found the pivot to be the first row's
    value all the way: 4e-5.  Pivots are now the mean of the first rows (k_column_stats: of each 32-row group; k_bn_finalize4: of the
    first counted panel): the module on its own (statistics from k_column_stats) and a dense layer's epilogue statistics (whose
    panel pivots are still first rows: looser) against float64."""
    from radargnn_amd import gnn, ops
    g = torch.Generator().manual_seed(3)
    h = torch.randn(m, 224, generator=g) * 0.2 + torch.linspace(-0.3, 0.3, 224)
    h[:4] += torch.randn(4, 224, generator=g).sign() * 20 * 0.2 * (1.0 + torch.rand(4, 224, generator=g))      # 20 - 40 spreads out
    bn = gnn.linear.BatchNorm(224).cuda()
    with torch.no_grad():
        bn.module.weight.uniform_(0.5, 1.5, generator=None); bn.module.bias.normal_(0, 0.3)
    want = _bn64(h, bn.module.weight, bn.module.bias)
    assert normwise(bn(h.cuda()), want) < 1e-6
    # the same rows out of a dense layer (statistics from the epilogue, per 128-row panel about the panel's first row)
    x = torch.randn(m, 64, generator=g)
    x[:4] *= 40.0
    w = torch.randn(224, 64, generator=g) * 0.1
    out, stats = ops.linear(x.cuda(), w.cuda(), None, want_stats=True)
    table = ops.batchnorm_finalize(stats, m, 224, bn.module.weight.detach(), bn.module.bias.detach(), None, None, None, True, 0.1, 1e-5)
    got = ops.scale_shift_act(out, table, relu=False)
    # (a panel whose first row is 40 spreads out still sums its 128 rows about that row in float32: 1e-5 here, 4e-5 before r05 moved the
    #  finalize's own pivot to the first panel's mean -- MEASUREMENTS.md section 8a lists the epilogue pivot as the open end)
    assert normwise(got, _bn64(out, bn.module.weight, bn.module.bias)) < 2e-5
