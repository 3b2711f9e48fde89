"""Window form of the max aggregation (rgnn_mpnn_aggregate_win, csrc/mpnn_tiles.hip) against a float64 evaluation of
gnn/mpnn_layers.py:94-101 + torch-scatter max in its hoisted form, M[t] = b + max_e(Q[s_e] + W_e a_e), and against the per-edge
kernel.  Tolerance: norm-wise 1e-5 per output tensor; the six-product bf16 split measures ~2e-7."""
import numpy as np
import pytest
import torch

from radargnn_amd import synthetic

pytestmark = pytest.mark.gpu


def _graph(kind, frames, **kw):
    from radargnn_amd import frames as fr
    from radargnn_amd.gnn.mpnn_layers import TargetCSR
    batch = fr.FrameBatch.from_frames(frames)
    if kind == "knn":
        g = fr.build_graphs(batch, fr.GraphSettings(algorithm="knn", k=kw["k"]))
        return g, TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order if kw.get("order", True) else None, all_sources=True)
    g = fr.build_graphs(batch, fr.GraphSettings(algorithm="radius", r=kw["r"]))
    return g, TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order if kw.get("order", True) else None, symmetric=True)


def _reference(Q, We, ea, bias, csr):
    n = csr.num_nodes
    rp = csr.rowptr.cpu().long()
    order = csr.order.cpu().long() if csr.order is not None else torch.arange(n)
    src = csr.src.cpu().long()
    Qc, eac = Q.cpu().double(), (ea.cpu().double() if ea is not None else None)
    Wc = We.cpu().double() if We is not None else None
    out = torch.zeros(n, Q.shape[1], dtype=torch.float64)
    has = torch.zeros(n, dtype=torch.bool)
    for p in range(n):
        lo, hi = int(rp[p]), int(rp[p + 1])
        if hi == lo:
            continue
        msg = Qc[src[lo:hi]]
        if eac is not None:
            msg = msg + eac[lo:hi] @ Wc.t()
        node = int(order[p])
        out[node] = msg.max(0).values + (bias.cpu().double() if bias is not None else 0.0)
        has[node] = True
    return out, has


# ---------------------------------------------------------------------------------------------------------------- window kernel
@pytest.mark.parametrize("kind,kw,d,de,with_bias,frames", [
    ("knn", dict(k=10), 144, 8, True, 6),
    ("knn", dict(k=20), 464, 8, True, 6),
    ("knn", dict(k=10, order=False), 50, 2, False, 6),     # no visiting order, channel count not a multiple of 32, 2 attributes
    ("radius", dict(r=6.0), 272, 8, True, 6),              # isolated nodes: empty segments
    ("radius", dict(r=2.5), 33, 5, True, 6),
    ("knn", dict(k=3), 32, 0, True, 6),                    # no edge attributes at all
    ("radius", dict(r=40.0), 96, 8, True, 3),              # crowded: in-degrees far above 64 -> the per-target kernel takes those targets
    ("knn", dict(k=60), 64, 8, True, 3),                   # 60 in-edges on average: streams hold one target, many go per target
])
def test_window_kernel_matches_float64(kind, kw, d, de, with_bias, frames):
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from radargnn_amd import ops
    fl = [synthetic.nuscenes_frame(i) for i in range(frames)]
    g, csr = _graph(kind, fl, **kw)
    n, e = csr.num_nodes, csr.num_edges
    assert e > 0
    gen = torch.Generator().manual_seed(d + de + 1)
    Q = ops.padded_rows(n, d, "cuda")
    Q.copy_((torch.randn(n, d, generator=gen) * 3.0).cuda())
    We = (torch.randn(d, de, generator=gen) * 0.5).cuda() if de else None
    ea = torch.randn(e, de, generator=gen).relu_().cuda() if de else None
    bias = torch.randn(d, generator=gen).cuda() if with_bias else None
    plan = ops.mpnn_win_plan(csr.rowptr, csr.src, csr.order)
    exp, has = _reference(Q, We, ea, bias, csr)
    for skip in (False, True):
        out = ops.mpnn_aggregate_win(bias, Q, We, ea, csr.rowptr, csr.src, plan, node_order=csr.order, skip_empty_rows=skip)
        got = out.cpu().double()
        err = ((got[has] - exp[has]).abs().max() / exp[has].abs().max()).item()
        assert err < 1e-5, err
        if not skip:
            assert bool((got[~has] == 0).all())
    again = ops.mpnn_aggregate_win(bias, Q, We, ea, csr.rowptr, csr.src, plan, node_order=csr.order, skip_empty_rows=True)
    assert torch.equal(again[has.cuda()], out[has.cuda()])          # (the plan's ticket counters were left at zero)
    # a rebuilt plan gives the same rows (the hash order of the distinct sources does not reach the results)
    plan2 = ops.mpnn_win_plan(csr.rowptr, csr.src, csr.order)
    third = ops.mpnn_aggregate_win(bias, Q, We, ea, csr.rowptr, csr.src, plan2, node_order=csr.order, skip_empty_rows=True)
    assert torch.equal(third[has.cuda()], out[has.cuda()])


def test_window_kernel_inside_the_model(monkeypatch):
    """DetNetBasic on a k = 20 batch big enough for TargetCSR.wants_window_kernel(): logits / boxes within 1e-5 of the float64 oracle
    with the window kernel, and the per-edge kernel within 1e-5 of the same oracle."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from oracle import gnn_oracle, graph_oracle
    from radargnn_amd import frames as fr, gnn
    from radargnn_amd.gnn import mpnn_layers
    frames = [synthetic.radarscenes_frame(i) for i in range(5)]           # 15 000 nodes, 300 000 edges
    settings = fr.GraphSettings(algorithm="knn", k=20)
    cfg = gnn.GNNArchitectureConfig(5, 2, [64, 32], [6], [16, 5], True, True, [32, 64], [4, 8, 16], "MPNNConv", False)
    torch.manual_seed(1)
    model = gnn.DetNetBasic(cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda()
    batch = fr.FrameBatch.from_frames(frames)
    outs = {}
    for use in (True, False):
        monkeypatch.setattr(mpnn_layers, "USE_WINDOW_KERNEL", use)
        cls, bb, g = fr.HotPath(model, settings)(batch)
        g.check()
        outs[use] = (cls.double().cpu(), bb.double().cpu())
        model.load_state_dict(sd)                                          # (same running statistics for both passes)
    graphs = [graph_oracle.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, "knn", 20, None, list(settings.node_features),
                                             list(settings.edge_features), "directed") for f in frames]
    ref = graph_oracle.collate(graphs)
    c64, b64 = gnn_oracle.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]), torch.from_numpy(ref["edge_attr"]),
                                        sd, dtype=torch.float64)
    for use in (True, False):
        assert ((outs[use][0] - c64).abs().max() / c64.abs().max()).item() < 1e-5
        assert ((outs[use][1] - b64).abs().max() / b64.abs().max()).item() < 1e-5
    assert not torch.equal(outs[True][0], outs[False][0])                  # (the two kernels differ in the last bits: both ran)
    # the plan built on the side stream (TargetCSR.start_win_plan: forked behind the CSR, joined by the first aggregation) gives the bits
    # of the plan built in line, eagerly, while capturing, and in replays of the captured step
    monkeypatch.setattr(mpnn_layers, "USE_WINDOW_KERNEL", True)
    for side in (True, False):
        monkeypatch.setattr(mpnn_layers, "PLAN_ON_SIDE_STREAM", side)
        hp = fr.HotPath(model, settings)
        for _ in range(4):                                                 # eager, capture, two replays (train mode: batch statistics)
            cls, bb, g = hp(batch)
            assert (g.csr is not None) and (getattr(g.csr, "_win_plan_pending", None) is None)
            assert torch.equal(cls.double().cpu(), outs[True][0]) and torch.equal(bb.double().cpu(), outs[True][1])


def test_window_kernel_on_a_radius_batch_inside_the_model(monkeypatch):
    """The r = 1 m graphs of the headline workload (4 edges per node; symmetric CSR built without the twin search, attributes of the
    in-edges = minus the own edges') take the window kernel from 2^18 edges on (TargetCSR.wants_window_kernel): logits / boxes of a
    24-frame batch within 1e-5 of the float64 oracle on both kernels, the plan built on the side stream behind the CSR, replays of the
    captured step bit-equal to the eager pass."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test but no GPU visible")
    from oracle import gnn_oracle, graph_oracle
    from radargnn_amd import frames as fr, gnn, ops
    from radargnn_amd.gnn import mpnn_layers
    frames = [synthetic.radarscenes_frame(i) for i in range(24)]
    monkeypatch.setattr(mpnn_layers, "WINDOW_KERNEL_MIN_EDGES_SPARSE", 1 << 18)     # (the rule starts at 2^19 edges on sparse graphs: 44 frames)
    settings = fr.GraphSettings(algorithm="radius", r=1.0)
    cfg = gnn.GNNArchitectureConfig(5, 2, [64, 32], [6], [16, 5], True, True, [32, 64], [4, 8, 16], "MPNNConv", False)
    torch.manual_seed(2)
    model = gnn.DetNetBasic(cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda()
    batch = fr.FrameBatch.from_frames(frames)
    outs = {}
    for use in (True, False):
        monkeypatch.setattr(mpnn_layers, "USE_WINDOW_KERNEL", use)
        before = ops.COUNTERS.get("mpnn_win", 0)
        hp = fr.HotPath(model, settings)
        cls, bb, g = hp(batch)
        g.check()
        assert g.edge_index.shape[1] >= (1 << 18)
        assert (ops.COUNTERS.get("mpnn_win", 0) - before == 2) == use          # two conv layers
        outs[use] = (cls.double().cpu(), bb.double().cpu())
        if use:
            for _ in range(3):                                                 # capture, two replays (train mode: batch statistics)
                c2, b2, _ = hp(batch)
                assert torch.equal(c2.double().cpu(), outs[True][0]) and torch.equal(b2.double().cpu(), outs[True][1])
        model.load_state_dict(sd)
    graphs = [graph_oracle.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, "radius", None, 1.0, list(settings.node_features),
                                             list(settings.edge_features), "directed") for f in frames]
    ref = graph_oracle.collate(graphs)
    c64, b64 = gnn_oracle.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]), torch.from_numpy(ref["edge_attr"]),
                                        sd, dtype=torch.float64)
    for use in (True, False):
        assert ((outs[use][0] - c64).abs().max() / c64.abs().max()).item() < 1e-5
        assert ((outs[use][1] - b64).abs().max() / b64.abs().max()).item() < 1e-5
    assert not torch.equal(outs[True][0], outs[False][0])
