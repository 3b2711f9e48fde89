"""Backward pass of the HIP path (gnn/trainer.py:176-231 trains through the modules with
``loss.backward()``) against autograd on the float64 CPU oracle (oracle/gnn_oracle.py is plain torch, hence
differentiable).

Tolerance: gradients are compared norm-wise per tensor, max|a - b| <= 2e-5 * max|b| -- the r02 path (no atomics, bf16x3
weight gradients, winners of the max aggregation recorded by the forward pass) measures 1e-7 ... 1e-6 on these tests, so the
bar sits an order above what is measured and an order below the 2e-4 it had while the edge stage still used float atomics.
Random continuous inputs: no ties at the max aggregation, no activations at exactly 0."""
import numpy as np
import pytest
import torch

from oracle import gnn_oracle as G

pytestmark = pytest.mark.gpu
GTOL = 2e-5


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
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def random_graph(n, e, seed, isolated=5):
    """Random directed edges without duplicates and without self loops; the last `isolated` nodes receive none."""
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (3 * e,), generator=g)
    dst = torch.randint(0, n - isolated, (3 * e,), generator=g)
    keep = src != dst
    pairs = torch.unique(torch.stack([src[keep], dst[keep]]), dim=1)
    perm = torch.randperm(pairs.shape[1], generator=g)[:e]
    return pairs[:, perm].contiguous()


def oracle_grads(model, x, ei, ea, conv_type, aggr, rc, rb):
    sd = {k: v.detach().cpu().double().requires_grad_(v.is_floating_point() and "running" not in k)
          if v.is_floating_point() else v.detach().cpu() for k, v in model.state_dict().items()}
    x64 = x.detach().cpu().double().requires_grad_(True)
    ea64 = ea.detach().cpu().double().requires_grad_(True)
    c, bb = G.det_net_basic(x64, ei.cpu(), ea64, sd, conv_layer_type=conv_type, aggr=aggr, training=True, dtype=torch.float64)
    loss = (c * rc.double()).sum() + (bb * rb.double()).sum()
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    return loss.item(), grads, x64.grad, ea64.grad, (c.detach(), bb.detach())


CASES = [
    # (conv type, aggr, edge encoder, node emb, edge emb, conv dims, batch-norm in mlps[, pre_layers, post_layers])
    ("MPNNConv", "max", False, [16, 24], [4, 8, 16], [24, 16], False),
    ("MPNNConv", "max", False, [16, 24], [4, 8, 16], [24, 16], False, 2, 2),     # deeper message / update MLPs
    ("MPNNConv", "mean", True, [8, 12], None, [12, 20], False, 3, 1),
    ("RadarPointGNNConv", "add", False, [16, 24], [4, 8], [24, 24], False, 2, 1),
    ("MPNNConv", "mean", False, [16, 24], [4, 8, 16], [24, 16], False),
    ("MPNNConv", "add", False, None, None, [12, 8], False),
    ("MPNNConv", "max", True, [8, 12], None, [12, 20], False),
    ("MPNNConv", "max", False, [16, 24], [4, 8], [24, 16], True),
    ("RadarPointGNNConv", "max", False, [16, 24], [4, 8, 16], [24, 24], False),
    ("RadarPointGNNConv", "mean", False, None, None, [5, 5], False),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in (c[0], c[1], c[2], c[6]) + tuple(c[7:])))
def test_det_net_backward_matches_float64_autograd(rg, case):
    conv_type, aggr, enc, node_emb, edge_emb, dims, bn_mlp = case[:7]
    pre_layers, post_layers = (case[7], case[8]) if len(case) > 7 else (1, 1)
    gnn, _ = rg
    torch.manual_seed(11)
    n, e, dn, de = 400, 2400, 5, 2
    cfg = gnn.GNNArchitectureConfig(
        node_feature_dimension=dn, edge_feature_dimension=de, conv_layer_dimensions=dims,
        classification_head_layer_dimensions=[6], regression_head_layer_dimensions=[16, 5],
        initial_node_feature_embedding=node_emb is not None, initial_edge_feature_embedding=edge_emb is not None,
        node_feature_embedding_layer_dimensions=node_emb or [], edge_feature_embedding_layer_dimensions=edge_emb or [],
        conv_layer_type=conv_type, batch_norm_in_mlps=bn_mlp, conv_use_edge_encoder=enc, aggregation_function=aggr,
        conv_pre_mlp_layer_number=pre_layers, conv_post_mlp_layer_number=post_layers)
    model = gnn.DetNetBasic(cfg).cuda()
    with torch.no_grad():                                        # non-trivial BatchNorm affine parameters
        for bn in model.batch_norms:
            bn.module.weight.uniform_(0.5, 1.5)
            bn.module.bias.uniform_(-0.5, 0.5)
    ei = random_graph(n, e, seed=5)
    x = torch.randn(n, dn)
    ea = torch.randn(ei.shape[1], de)
    rc, rb = torch.randn(n, 6), torch.randn(n, 5)
    exp_loss, exp_g, exp_dx, exp_dea, (exp_c, exp_bb) = oracle_grads(model, x, ei, ea, conv_type, aggr, rc, rb)

    xg = x.cuda().requires_grad_(True)
    eag = ea.cuda().requires_grad_(True)
    rm_before = model.batch_norms[0].module.running_mean.clone()
    c, bb = model(xg, ei.cuda(), eag)
    assert c.requires_grad and bb.requires_grad
    assert normwise(c, exp_c) < 1e-5 and normwise(bb, exp_bb) < 1e-5
    rm_after_fwd = model.batch_norms[0].module.running_mean.clone()
    loss = (c * rc.cuda()).sum() + (bb * rb.cuda()).sum()
    loss.backward()
    # the re-execution inside backward must not update the running statistics a second time
    assert torch.equal(model.batch_norms[0].module.running_mean, rm_after_fwd)
    assert not torch.equal(rm_after_fwd, rm_before)
    assert abs(loss.item() - exp_loss) <= 1e-4 * max(1.0, abs(exp_loss))
    worst = {}
    # a bias in front of a train-mode BatchNorm has an exactly-zero gradient (the mean is subtracted again): its
    # float64 value is ~1e-17, so the error (fp32 cancellation noise of a sum of ~10^5 terms) is measured against 5 % of
    # the largest gradient instead
    largest = max(float(v.abs().max()) for v in exp_g.values())
    floor = 5e-2 * largest
    for name, p in model.named_parameters():
        assert p.grad is not None, f"{name} got no gradient"
        ref = exp_g[name]
        err = float((p.grad.detach().double().cpu() - ref).abs().max())
        # (an exactly-zero gradient -- float64 value below 1e-9 of the largest one -- is pure cancellation noise: against the
        #  largest gradient of the model; everything else against its own magnitude, floored at 5 % of the largest)
        zero_grad = float(ref.abs().max()) < 1e-9 * largest
        worst[name] = err / (largest if zero_grad else max(float(ref.abs().max()), floor))
    bad = {k: v for k, v in worst.items() if not v < GTOL}
    assert not bad, bad
    assert normwise(xg.grad, exp_dx) < GTOL
    assert normwise(eag.grad, exp_dea) < GTOL


def test_training_in_the_f16x2_form_with_every_bound_verified(rg, monkeypatch):
    """ops.TRAIN_F16X2: a recorded forward tracks bounds like the inference one and its backward goes on in the same pool -- max |M|
    from rgnn_mpnn_aggregate_max_arg_absmax, max |dh| from rgnn_bn_bwd_apply_absmax, max |dQ| from rgnn_mpnn_max_bwd_absmax, the
    dense launches' own outputs -- so forward AND dgrad launches take the f16x2 form.  Widths above the 32-column floor of the
    split-operand kernels; RGNN_CHECK_BOUNDS compares every bound a launch consumes with the operand it describes (also on
    autograd's thread); gradients against float64 autograd at the tolerance of the bf16x3 path."""
    gnn, ops = rg
    monkeypatch.setattr(ops, "TRAIN_F16X2", True)
    monkeypatch.setattr(ops, "CHECK_BOUNDS", True)
    torch.manual_seed(13)
    # (small enough that no near-tie at a max aggregation is resolved differently by fp32 and float64 -- 112 000 (target, channel)
    #  maxima; at 624 000 one flips, which moves a channel's gradient to another source: 1e-2 of the largest gradient)
    n, e, dn, de = 1000, 5000, 5, 2
    cfg = gnn.GNNArchitectureConfig(
        node_feature_dimension=dn, edge_feature_dimension=de, conv_layer_dimensions=[64, 48],
        classification_head_layer_dimensions=[6], regression_head_layer_dimensions=[16, 5],
        initial_node_feature_embedding=True, initial_edge_feature_embedding=True,
        node_feature_embedding_layer_dimensions=[32, 64], edge_feature_embedding_layer_dimensions=[4, 8, 16],
        conv_layer_type="MPNNConv", batch_norm_in_mlps=False, conv_use_edge_encoder=False, aggregation_function="max",
        conv_pre_mlp_layer_number=1, conv_post_mlp_layer_number=1)
    model = gnn.DetNetBasic(cfg).cuda()
    with torch.no_grad():
        for bn in model.batch_norms:
            bn.module.weight.uniform_(0.5, 1.5)
            bn.module.bias.uniform_(-0.5, 0.5)
    ei = random_graph(n, e, seed=6, isolated=40)
    x = torch.randn(n, dn)
    ea = torch.randn(ei.shape[1], de)
    rc, rb = torch.randn(n, 6), torch.randn(n, 5)
    exp_loss, exp_g, exp_dx, exp_dea, (exp_c, exp_bb) = oracle_grads(model, x, ei, ea, "MPNNConv", "max", rc, rb)
    results = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "TRAIN_F16X2", on)
        model.zero_grad()
        xg = x.cuda().requires_grad_(True)
        eag = ea.cuda().requires_grad_(True)
        before = ops.COUNTERS.get("f16x2", 0)
        c, bb = model(xg, ei.cuda(), eag)
        fwd = ops.COUNTERS.get("f16x2", 0) - before
        ((c * rc.cuda()).sum() + (bb * rb.cuda()).sum()).backward()
        torch.cuda.synchronize()
        total = ops.COUNTERS.get("f16x2", 0) - before
        results[on] = (fwd, total - fwd)
        assert normwise(c, exp_c) < 1e-5 and normwise(bb, exp_bb) < 1e-5
        largest = max(float(v.abs().max()) for v in exp_g.values())
        for name, p in model.named_parameters():
            ref = exp_g[name]
            err = float((p.grad.detach().double().cpu() - ref).abs().max())
            zero_grad = float(ref.abs().max()) < 1e-9 * largest
            assert err / (largest if zero_grad else max(float(ref.abs().max()), 5e-2 * largest)) < GTOL, (name, on)
        assert normwise(xg.grad, exp_dx) < GTOL and normwise(eag.grad, exp_dea) < GTOL
    assert results[True][0] >= 4 and results[True][1] >= 4, results       # conv layers' forward AND dgrad launches took the form
    assert results[False] == (0, 0), results


@pytest.mark.parametrize("conv,aggr,pre,post,dims", [("MPNNConv", "add", 2, 2, [64, 48]), ("RadarPointGNNConv", "max", 1, 1, [64, 64]),
                                                     ("MPNNConv", "max", 1, 2, [64, 48]), ("RadarPointGNNConv", "add", 2, 1, [64, 64])])
def test_training_bounds_survive_summed_gradients_on_unfolded_layers(rg, monkeypatch, conv, aggr, pre, post, dims):
    """in the forms that are NOT one autograd node per layer -- deeper message / update MLPs, add aggregation,
    RadarPointGNNConv (x is also the residual) -- a layer's input has several consumers, autograd's InputBuffer sums their gradients
    IN PLACE into the tensor that arrived first and hands that object on, `_rgnn_bound` attribute included.  Bounds are stamped with
    the tensor's version (ops.set_bound / bound_of), so the sum carries no bound and its consumers leave the f16x2 form instead of
    pre-scaling by one contributor's maximum.  RGNN_CHECK_BOUNDS verifies every bound that IS consumed; gradients against float64."""
    gnn, ops = rg
    monkeypatch.setattr(ops, "TRAIN_F16X2", True)
    monkeypatch.setattr(ops, "CHECK_BOUNDS", True)
    torch.manual_seed(17)
    n, e, dn, de = 900, 4000, 5, 2
    cfg = gnn.GNNArchitectureConfig(
        node_feature_dimension=dn, edge_feature_dimension=de, conv_layer_dimensions=dims,
        classification_head_layer_dimensions=[6], regression_head_layer_dimensions=[16, 5],
        initial_node_feature_embedding=True, initial_edge_feature_embedding=True,
        node_feature_embedding_layer_dimensions=[32, 64], edge_feature_embedding_layer_dimensions=[4, 8, 16],
        conv_layer_type=conv, batch_norm_in_mlps=False, conv_use_edge_encoder=False, aggregation_function=aggr,
        conv_pre_mlp_layer_number=pre, conv_post_mlp_layer_number=post)
    model = gnn.DetNetBasic(cfg).cuda()
    ei = random_graph(n, e, seed=9, isolated=30)
    x = torch.randn(n, dn)
    ea = torch.randn(ei.shape[1], de)
    rc, rb = torch.randn(n, 6), torch.randn(n, 5)
    # (a gradient 1000x the forward's scale: a stale bound from ONE contributor of a sum would be far too small for the sum)
    rc, rb = rc * 1e3, rb * 1e3
    exp_loss, exp_g, exp_dx, exp_dea, (exp_c, exp_bb) = oracle_grads(model, x, ei, ea, conv, aggr, rc, rb)
    xg = x.cuda().requires_grad_(True)
    eag = ea.cuda().requires_grad_(True)
    c, bb = model(xg, ei.cuda(), eag)
    ((c * rc.cuda()).sum() + (bb * rb.cuda()).sum()).backward()
    torch.cuda.synchronize()
    assert normwise(c, exp_c) < 1e-5 and normwise(bb, exp_bb) < 1e-5
    largest = max(float(v.abs().max()) for v in exp_g.values())
    for name, p in model.named_parameters():
        ref = exp_g[name]
        assert torch.isfinite(p.grad).all(), name
        err = float((p.grad.detach().double().cpu() - ref).abs().max())
        zero_grad = float(ref.abs().max()) < 1e-9 * largest
        assert err / (largest if zero_grad else max(float(ref.abs().max()), 5e-2 * largest)) < GTOL, name
    assert normwise(xg.grad, exp_dx) < GTOL and normwise(eag.grad, exp_dea) < GTOL


def test_a_bound_does_not_survive_an_in_place_write_through_torch(rg):
    """The mechanism of the test above in isolation: ops.bound_of returns the word while the tensor is as the kernel left it and
    None once torch wrote it in place (version counter), e.g. autograd's gradient accumulation."""
    gnn, ops = rg
    t = torch.ones(4, 4, device="cuda")
    w = torch.zeros(ops.BOUND_SLOTS, device="cuda")
    ops.set_bound(t, w)
    assert ops.bound_of(t) is w
    t.add_(5.0)
    assert ops.bound_of(t) is None
    ops.set_bound(t, w)
    assert ops.bound_of(t) is w
    ops.set_bound(t, None)
    assert ops.bound_of(t) is None


@pytest.mark.parametrize("m,k1,k2,n", [(1024, 32, 0, 64), (5000, 224, 464, 224), (4097, 128, 0, 544), (3000, 64, 272, 68),
                                       (20000, 224, 0, 464), (6001, 5, 0, 32), (777, 3, 7, 6), (15, 16, 0, 16),
                                       (50001, 8, 0, 16), (9000, 16, 0, 5), (800, 2, 0, 4)])   # (the last four: k_wgrad_narrow)
def test_weight_gradient_kernel_matches_float64(rg, m, k1, k2, n):
    """dW = G^T [A1 | A2 | 1] on the bf16x3 MFMA kernel (row slabs, partial tiles, deterministic reduction): any widths and
    strides (the 5-wide raw node features included), bias gradient as the column of ones, as accurate as the fp32-MFMA
    kernel against float64."""
    _, ops = rg
    g_ = torch.Generator().manual_seed(m + n)
    G_ = torch.randn(m, n, generator=g_)
    A1 = torch.randn(m, k1, generator=g_)
    A2 = torch.randn(m, k2, generator=g_) if k2 else None
    A = A1 if A2 is None else torch.cat([A1, A2], 1)
    exp = G_.double().t() @ torch.cat([A.double(), torch.ones(m, 1, dtype=torch.float64)], 1)
    wide = torch.randn(m, k1 + 8, generator=g_).cuda()            # a1 as a column view of a wider matrix (row stride > width)
    wide[:, :k1] = A1.cuda()
    args = (G_.cuda(), wide[:, :k1], None if A2 is None else A2.cuda())
    assert ops.linear_wgrad_supported(*args)
    got = ops.linear_wgrad(*args, with_bias=True)
    assert got.shape == (n, k1 + k2 + 1)
    assert normwise(got, exp) < 2e-6
    assert torch.equal(got, ops.linear_wgrad(*args, with_bias=True))    # no atomics: bit-identical on repetition
    plain = ops.linear_wgrad(*args)
    assert plain.shape == (n, k1 + k2) and torch.equal(plain, got[:, :-1])
    if n % 4 == 0 and k1 % 4 == 0 and k2 % 4 == 0 and m >= 1024:
        assert normwise(plain, exp[:, :-1]) < 4 * normwise(ops.linear_wgrad_fp32(*args), exp[:, :-1]) + 2e-7


def test_batchnorm_backward_mask_from_the_apply_table_gives_the_bits_of_the_mask_from_y(rg, monkeypatch):
    """ops.BN_BWD_MASK_FROM_TABLE: relu'(y) recomputed as fmaf(h - mean_hi, g, t) > 0 from the BatchNorm input and the forward pass's
    apply table -- the arithmetic that produced y -- instead of reading y: every gradient bit-identical, train and eval mode."""
    gnn, ops = rg
    torch.manual_seed(21)
    n, e = 1500, 9000
    cfg = gnn.GNNArchitectureConfig(5, 2, [48, 40], [6], [16, 5], True, True, [16, 32], [4, 8, 16], "MPNNConv", True)
    model = gnn.DetNetBasic(cfg).cuda()
    ei = random_graph(n, e, seed=8).cuda()
    x, ea = torch.randn(n, 5).cuda(), torch.randn(ei.shape[1], 2).cuda()
    rc, rb = torch.randn(n, 6).cuda(), torch.randn(n, 5).cuda()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for training in (True, False):
        model.train(training)
        got = {}
        for from_table in (True, False):
            monkeypatch.setattr(ops, "BN_BWD_MASK_FROM_TABLE", from_table)
            model.load_state_dict(sd)
            model.zero_grad()
            xg, eag = x.clone().requires_grad_(True), ea.clone().requires_grad_(True)
            c, bb = model(xg, ei, eag)
            ((c * rc).sum() + (bb * rb).sum()).backward()
            got[from_table] = [xg.grad.clone(), eag.grad.clone()] + [p.grad.clone() for p in model.parameters()]
        assert all(torch.equal(a, b) for a, b in zip(got[True], got[False])), training


@pytest.mark.parametrize("m,k1,k2,n,sg,s1,s2", [(5000, 224, 464, 224, 1.0, 1.0, 1.0), (20000, 224, 0, 464, 1e-4, 30.0, 1.0),
                                                 (4097, 128, 272, 544, 3e3, 1e-3, 1e-3), (3000, 64, 272, 68, 1e-6, 1e4, 2e2),
                                                 (9000, 5, 0, 48, 1.0, 1e-5, 1.0), (1500, 96, 40, 100, 1e12, 1e-12, 1e-9)])
def test_weight_gradient_in_the_f16x2_form_matches_float64(rg, m, k1, k2, n, sg, s1, s2):
    """rgnn_wgrad_bounds: both operands pre-scaled by powers of two from their bounds, two f16 terms, three products -- against float64
    on operands of very different magnitudes (the column of ones travels as 2^14 after the pre-scale whatever the scale of A is),
    blocks of A a few decades apart, over all rows and over a row list; the error is measured against bound(G) bound(A) like the
    forward f16x2 form's (tests/test_gpu_f16x2.py), and next to the bf16x3 form on the same data."""
    _, ops = rg
    g_ = torch.Generator().manual_seed(m + n + k1)
    G_ = torch.randn(m, n, generator=g_) * sg
    A1 = torch.randn(m, k1, generator=g_) * s1
    A2 = torch.randn(m, k2, generator=g_) * s2 if k2 else None
    A = A1 if A2 is None else torch.cat([A1, A2], 1)
    full = torch.cat([A.double(), torch.ones(m, 1, dtype=torch.float64)], 1)
    Gc, A1c, A2c = G_.cuda(), A1.cuda(), None if A2 is None else A2.cuda()
    rows = torch.randperm(m, generator=g_)[: m // 2 + 1]
    lst = torch.full((m,), -7, dtype=torch.int32)
    lst[:rows.numel()] = rows.int()
    cnt = torch.tensor([rows.numel()], dtype=torch.int64).cuda()
    before = dict(ops.COUNTERS)
    with ops.using_bounds(ops.BoundPool("cuda")):
        bounds = (ops.make_bound(Gc.abs().max()), ops.make_bound(A1c.abs().max()), None if A2c is None else ops.make_bound(A2c.abs().max()))
        got = ops.linear_wgrad(Gc, A1c, A2c, with_bias=True, bounds=bounds)
        again = ops.linear_wgrad(Gc, A1c, A2c, with_bias=True, bounds=bounds)
        part = ops.linear_wgrad(Gc, A1c, A2c, with_bias=True, row_index=lst.cuda(), m_dev=cnt, bounds=bounds)
        x3 = ops.linear_wgrad(Gc, A1c, A2c, with_bias=True, bounds=(None, None, None))
    assert ops.COUNTERS.get("wgrad_f16x2", 0) - before.get("wgrad_f16x2", 0) == 3
    assert torch.equal(got, again)
    for got_, sel in ((got, slice(None)), (part, rows)):
        exp = G_[sel].double().t() @ full[sel]
        # column blocks separately: every block against ITS largest entry (blocks of A differ by decades, and so do their gradients)
        for lo, hi in ((0, k1), (k1, k1 + k2), (k1 + k2, k1 + k2 + 1)):
            if hi > lo:
                assert normwise(got_[:, lo:hi], exp[:, lo:hi]) < 2e-5, (lo, hi, normwise(got_[:, lo:hi], exp[:, lo:hi]))
    exp = G_.double().t() @ full
    assert normwise(got, exp) < 4 * normwise(x3, exp) + 2e-6


def test_weight_gradient_over_a_row_list(rg):
    """Rows taken through a device-side row list with a device-side count (the targets with / without incoming edges)."""
    _, ops = rg
    g_ = torch.Generator().manual_seed(9)
    m, n, k1, k2 = 9000, 96, 64, 40
    G_, A1, A2 = torch.randn(m, n, generator=g_), torch.randn(m, k1, generator=g_), torch.randn(m, k2, generator=g_)
    rows = torch.randperm(m, generator=g_)[:5003]
    lst = torch.full((m,), -7, dtype=torch.int32)
    lst[:rows.numel()] = rows.int()
    cnt = torch.tensor([rows.numel()], dtype=torch.int64)
    exp = G_[rows].double().t() @ torch.cat([A1[rows].double(), A2[rows].double(), torch.ones(rows.numel(), 1, dtype=torch.float64)], 1)
    got = ops.linear_wgrad(G_.cuda(), A1.cuda(), A2.cuda(), with_bias=True, row_index=lst.cuda(), m_dev=cnt.cuda())
    assert normwise(got, exp) < 2e-6
    empty = ops.linear_wgrad(G_.cuda(), A1.cuda(), A2.cuda(), with_bias=True, row_index=lst.cuda(),
                             m_dev=torch.zeros(1, dtype=torch.int64).cuda())
    assert float(empty.abs().max()) == 0.0


def test_standalone_layers_backward(rg):
    gnn, _ = rg
    torch.manual_seed(3)
    for conv_type in ("MPNNConv", "RadarPointGNNConv"):
        n, e, c_in, de = 300, 1500, 12, 3
        ei = random_graph(n, e, seed=9)
        if conv_type == "MPNNConv":
            conv = gnn.MPNNConv(c_in, 20, de, aggr="max").cuda()
            ofn = G.mpnn_conv
        else:
            conv = gnn.RadarPointGNNConv(c_in, de, aggr="max").cuda()
            ofn = G.radar_point_gnn_conv
        x, ea = torch.randn(n, c_in), torch.randn(ei.shape[1], de)
        r = torch.randn(n, conv.out_channels)
        sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in conv.state_dict().items()}
        x64, ea64 = x.double().requires_grad_(True), ea.double().requires_grad_(True)
        out64 = ofn(x64, ei, ea64, sd, "", "max")
        (out64 * r.double()).sum().backward()
        xg, eag = x.cuda().requires_grad_(True), ea.cuda().requires_grad_(True)
        out = conv(xg, ei.cuda(), eag)
        assert normwise(out, out64) < 1e-5
        (out * r.cuda()).sum().backward()
        for name, p in conv.named_parameters():
            assert normwise(p.grad, sd[name].grad) < GTOL, name
        assert normwise(xg.grad, x64.grad) < GTOL and normwise(eag.grad, ea64.grad) < GTOL


def test_linear_and_batchnorm_modules_backward(rg):
    gnn, _ = rg
    torch.manual_seed(0)
    mlp = gnn.get_mlp(7, 4, [16, 12], True).cuda()
    ref = torch.nn.Sequential(torch.nn.Linear(7, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Linear(16, 12),
                              torch.nn.BatchNorm1d(12), torch.nn.ReLU(), torch.nn.Linear(12, 4)).double()
    with torch.no_grad():
        for i in (0, 3, 6):
            ref[i].weight.copy_(mlp[i].weight.double().cpu()); ref[i].bias.copy_(mlp[i].bias.double().cpu())
    x = torch.randn(500, 7)
    r = torch.randn(500, 4)
    xg = x.cuda().requires_grad_(True)
    y = mlp(xg)                                                  # nn.Sequential.__call__: module by module
    (y * r.cuda()).sum().backward()
    x64 = x.double().requires_grad_(True)
    y64 = ref(x64)
    (y64 * r.double()).sum().backward()
    assert normwise(y, y64) < 1e-5
    assert normwise(xg.grad, x64.grad) < GTOL
    for i in (0, 3, 6):
        assert normwise(mlp[i].weight.grad, ref[i].weight.grad) < GTOL
        if i == 6:
            assert normwise(mlp[i].bias.grad, ref[i].bias.grad) < GTOL
        else:   # bias in front of a train-mode BatchNorm: the exact gradient is 0, measure against the weight gradient
            assert float(mlp[i].bias.grad.abs().max()) < GTOL * float(ref[i].weight.grad.abs().max())
    for i in (1, 4):
        assert normwise(mlp[i].module.weight.grad, ref[i].weight.grad) < GTOL
        assert normwise(mlp[i].module.bias.grad, ref[i].bias.grad) < GTOL


def test_backward_in_eval_mode_uses_running_statistics(rg):
    """model.eval(): BatchNorm normalises with the running statistics and its backward is a plain per-column scale.
    (Seed: the max aggregation must have no NEAR-tie either -- two messages of a target within fp32 rounding of each other;
    float64 and fp32 then pick different winners and the gradient of that one element goes to another source.  Seed 7 has
    one such pair among 18 176 in the second conv, gap < 1e-6 relative.)"""
    gnn, _ = rg
    torch.manual_seed(11)
    cfg = gnn.GNNArchitectureConfig(5, 2, [24, 16], [6], [16, 5], True, True, [16, 24], [4, 8, 16], "MPNNConv", True)
    model = gnn.DetNetBasic(cfg).cuda()
    n = 300
    ei = random_graph(n, 1500, seed=3)
    x, ea = torch.randn(n, 5), torch.randn(ei.shape[1], 2)
    with torch.no_grad():
        for _ in range(3):                                       # move the running statistics away from (0, 1)
            model(x.cuda(), ei.cuda(), ea.cuda())
    model.eval()
    rc, rb = torch.randn(n, 6), torch.randn(n, 5)
    sd = {k: (v.detach().cpu().double().requires_grad_("running" not in k) if v.is_floating_point() else v.detach().cpu())
          for k, v in model.state_dict().items()}
    x64, ea64 = x.double().requires_grad_(True), ea.double().requires_grad_(True)
    c64, b64 = G.det_net_basic(x64, ei, ea64, sd, training=False, dtype=torch.float64)
    ((c64 * rc.double()).sum() + (b64 * rb.double()).sum()).backward()
    rm = model.batch_norms[0].module.running_mean.clone()
    xg, eag = x.cuda().requires_grad_(True), ea.cuda().requires_grad_(True)
    c, b = model(xg, ei.cuda(), eag)
    assert normwise(c, c64) < 1e-5 and normwise(b, b64) < 1e-5
    ((c * rc.cuda()).sum() + (b * rb.cuda()).sum()).backward()
    assert torch.equal(model.batch_norms[0].module.running_mean, rm)          # eval mode: statistics untouched
    for name, p in model.named_parameters():
        assert normwise(p.grad, sd[name].grad) < GTOL, name
    assert normwise(xg.grad, x64.grad) < GTOL and normwise(eag.grad, ea64.grad) < GTOL


def test_inference_with_autograd_enabled_costs_no_graph_and_matches_no_grad(rg):
    """postprocessor/inference.py:57-62 calls the model with autograd enabled and never calls backward."""
    gnn, _ = rg
    torch.manual_seed(1)
    cfg = gnn.GNNArchitectureConfig(5, 2, [24, 16], [6], [16, 5], True, True, [16, 24], [4, 8, 16])
    model = gnn.DetNetBasic(cfg).cuda()
    ei = random_graph(300, 1500, seed=2).cuda()
    x, ea = torch.randn(300, 5).cuda(), torch.randn(ei.shape[1], 2).cuda()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    c1, bb1 = model(x, ei, ea)
    model.load_state_dict(sd0)
    with torch.no_grad():
        c2, bb2 = model(x, ei, ea)
    assert torch.equal(c1.detach(), c2) and torch.equal(bb1.detach(), bb2)
    assert c1.grad_fn is not None and c2.grad_fn is None
    assert type(c1.grad_fn).__name__.startswith("_Checkpointed")


def test_a_few_training_steps_reduce_the_loss(rg):
    """The trainer's step (trainer.py:175-231): zero_grad, forward, cross-entropy + Huber, backward, Adam."""
    gnn, _ = rg
    torch.manual_seed(4)
    cfg = gnn.GNNArchitectureConfig(5, 2, [32, 32], [6], [16, 5], True, True, [16, 32], [4, 8, 16])
    model = gnn.DetNetBasic(cfg).cuda()
    n = 600
    ei = random_graph(n, 3600, seed=8).cuda()
    x, ea = torch.randn(n, 5).cuda(), torch.randn(ei.shape[1], 2).cuda()
    label = torch.randint(0, 6, (n,)).cuda()
    box = torch.randn(n, 5).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        x.requires_grad_(); ea.requires_grad_()
        c, bb = model(x, ei, ea)
        loss = torch.nn.functional.cross_entropy(c, label) + torch.nn.functional.huber_loss(bb, box)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.8 * losses[0], losses


@pytest.mark.parametrize("de,pre,aggr", [(24, 1, "max"), (40, 1, "mean"), (40, 2, "add"), (20, 1, "max")])
def test_edge_attributes_wider_than_the_fused_kernels_take(rg, de, pre, aggr):
    """The fused message kernels take 32 edge attributes (their backward 16); a wider edge feature vector -- nothing the reference
    ships, but nothing it forbids either (gnn/mpnn_layers.py:64-68) -- goes through per-edge rows + a dense launch + the segmented
    reduce, forward and backward, with and without autograd."""
    gnn, _ = rg
    torch.manual_seed(5)
    n, e, c_in = 300, 1500, 12
    ei = random_graph(n, e, seed=4)
    conv = gnn.MPNNConv(c_in, 20, de, aggr=aggr, pre_layers=pre).cuda()
    x, ea = torch.randn(n, c_in), torch.randn(ei.shape[1], de)
    r = torch.randn(n, conv.out_channels)
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in conv.state_dict().items()}
    x64, ea64 = x.double().requires_grad_(True), ea.double().requires_grad_(True)
    out64 = G.mpnn_conv(x64, ei, ea64, sd, "", aggr)
    (out64 * r.double()).sum().backward()
    with torch.no_grad():
        assert normwise(conv(x.cuda(), ei.cuda(), ea.cuda()), out64) < 1e-5
    xg, eag = x.cuda().requires_grad_(True), ea.cuda().requires_grad_(True)
    out = conv(xg, ei.cuda(), eag)
    assert normwise(out, out64) < 1e-5
    (out * r.cuda()).sum().backward()
    for name, p in conv.named_parameters():
        assert normwise(p.grad, sd[name].grad) < GTOL, name
    assert normwise(xg.grad, x64.grad) < GTOL and normwise(eag.grad, ea64.grad) < GTOL


def test_backward_entry_points_on_empty_inputs(rg):
    """The C entry points themselves on a graph without edges / a matrix without rows (what the Python layer no longer forwards):
    rgnn_wgrad with m = 0 is the sum over nothing, rgnn_mpnn_aggregate_bwd with no edges leaves zero gradients."""
    import ctypes as C
    from radargnn_amd._lib import lib
    dW = torch.full((6, 9), 7.0, device="cuda")
    part = torch.empty(16, device="cuda")
    g = torch.empty((0, 6), device="cuda"); a = torch.empty((0, 8), device="cuda")
    rc = lib.rgnn_wgrad(None, 6, 6, None, 8, 8, None, 0, 0, 1, 0, None, None, part.data_ptr(), dW.data_ptr(), None)
    assert rc == 0 and float(dW.abs().max()) == 0.0
    n, d, de = 5, 8, 2
    dQ = torch.full((n, d), 3.0, device="cuda"); dWe = torch.full((d, de), 3.0, device="cuda")
    z32 = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    dM = torch.randn(n, d, device="cuda"); Q = torch.randn(n, d, device="cuda")
    rc = lib.rgnn_mpnn_aggregate_bwd(dM.data_ptr(), d, Q.data_ptr(), d, None, de, None, de, z32.data_ptr(), z32.data_ptr(), None, n, d, 2,
                                     z32.data_ptr(), z32.data_ptr(), z32.data_ptr(), None, 0, None, None, None, dQ.data_ptr(), d, None,
                                     dWe.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0 and float(dQ.abs().max()) == 0.0 and float(dWe.abs().max()) == 0.0
