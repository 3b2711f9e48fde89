"""Pins of the GNN oracle (oracle/gnn_oracle.py), CPU only:
* the five known answers of the reference's test/test_gnn.py (all-ones weights: the only GNN results the reference pins);
* hand-derived answers with ASYMMETRIC weights (tests/gnn_hand_vectors.py: the derivation is written out there) -- these
  distinguish [x_i | x_j | e] from [x_j | x_i | e], cover mean / add, a target without incoming edges, a duplicate edge and
  RadarPointGNNConv's residual;
* an independent float64 numpy restatement (oracle/gnn_numpy.py, one Python iteration per edge) on random graphs, incl.
  DetNetBasic end to end with train-mode BatchNorm, batch_norm_in_mlps and both conv types;
* if torch_geometric happens to be importable: the reference's own MPNNConv / RadarPointGNNConv / DetNetBasic built from
  /root/reference/src (self-skipping here: no wheel, no network)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import gnn_numpy as GN
from oracle import gnn_oracle as G

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gnn_hand_vectors as HV  # noqa: E402


def ones_linear(sd, key, out_f, in_f, value=1.0):
    sd[key + ".weight"] = torch.full((out_f, in_f), value)
    sd[key + ".bias"] = torch.zeros(out_f)


def test_get_mlp_known_answer():
    # test_gnn.py:9-25: get_mlp(2, 3, [5], False), all-ones weights, x = [1, 1] -> [10, 10, 10]
    sd = {}
    ones_linear(sd, "mlp.0", 5, 2)
    ones_linear(sd, "mlp.2", 3, 5)
    y = G.run_sequential(torch.tensor([[1.0, 1.0]]), sd, "mlp.")
    assert y.tolist() == [[10.0, 10.0, 10.0]]


def test_mpnn_conv_mlps_known_answer():
    # test_gnn.py:79-116: MPNNConv(2, 4, 3, post_layers=2): pre Linear(7,7); post Linear(9,4), ReLU, Linear(4,4)
    sd = {}
    ones_linear(sd, "c.pre_mlp.0", 7, 7)
    ones_linear(sd, "c.post_mlp.0", 4, 9)
    ones_linear(sd, "c.post_mlp.2", 4, 4)
    pre = G.run_sequential(torch.tensor([[1.0] * 7, [2.0] * 7]), sd, "c.pre_mlp.")
    post = G.run_sequential(torch.tensor([[1.0] * 9, [2.0] * 9]), sd, "c.post_mlp.")
    assert pre[0].tolist() == [7.0] * 7
    assert post[1].tolist() == [72.0] * 4


def test_mpnn_conv_forward_known_answer():
    # test_gnn.py:119-172: 2 nodes, 3 edges (one duplicated pair), max aggregation -> 436 at node 1
    sd = {}
    ones_linear(sd, "c.pre_mlp.0", 7, 7)
    ones_linear(sd, "c.post_mlp.0", 4, 9)
    ones_linear(sd, "c.post_mlp.2", 4, 4)
    x = torch.tensor([[1.0, 1.0], [2.0, 2.0]])
    ei = torch.tensor([[0, 1, 0], [1, 0, 1]])
    ea = torch.tensor([[3.0] * 3, [4.0] * 3, [1.0] * 3])
    out = G.mpnn_conv(x, ei, ea, sd, "c.", "max")
    assert out[1].tolist() == [436.0] * 4


def test_mpnn_conv_edge_encoder_known_answer():
    # test_gnn.py:175-221: MPNNConv(1, 4, 2, use_edge_encoder=True) -> 23
    sd = {}
    ones_linear(sd, "c.edge_encoder", 1, 2, 2.0)
    ones_linear(sd, "c.pre_mlp.0", 3, 3)
    ones_linear(sd, "c.post_mlp.0", 4, 4)
    x = torch.tensor([[1.0], [2.0]])
    ei = torch.tensor([[0, 1], [1, 0]])
    ea = torch.tensor([[1.0, 1.0], [2.0, 2.0]])
    out = G.mpnn_conv(x, ei, ea, sd, "c.", "max")
    assert out[1, 0].item() == 23.0


def test_scatter_semantics():
    msg = torch.tensor([[1.0, -2.0], [3.0, -4.0], [-5.0, -6.0]])
    idx = torch.tensor([2, 2, 0])
    assert G.scatter_reduce_rows(msg, idx, 4, "max").tolist() == [[-5, -6], [0, 0], [3, -2], [0, 0]]
    assert G.scatter_reduce_rows(msg, idx, 4, "add").tolist() == [[-5, -6], [0, 0], [4, -6], [0, 0]]
    assert G.scatter_reduce_rows(msg, idx, 4, "mean").tolist() == [[-5, -6], [0, 0], [2, -3], [0, 0]]


# ------------------------------------------------------------------------------------------------ hand-derived vectors
@pytest.mark.parametrize("case", HV.CASES, ids=lambda c: c["name"])
def test_hand_derived_vectors_torch_oracle(case):
    sd = {k: torch.tensor(v, dtype=torch.float64) for k, v in case["state_dict"].items()}
    x, ei, ea = (torch.tensor(case[k], dtype=torch.float64 if k != "edge_index" else torch.int64) for k in ("x", "edge_index", "edge_attr"))
    fn = G.mpnn_conv if case["kind"] == "MPNNConv" else G.radar_point_gnn_conv
    out = fn(x, ei, ea, sd, "c.", case["aggr"])
    np.testing.assert_allclose(out.numpy(), np.array(case["expected"]), rtol=0, atol=1e-9)


@pytest.mark.parametrize("case", HV.CASES, ids=lambda c: c["name"])
def test_hand_derived_vectors_numpy_oracle(case):
    sd = {k: np.array(v, dtype=np.float64) for k, v in case["state_dict"].items()}
    out = GN.conv(np.array(case["x"], dtype=np.float64), np.array(case["edge_index"]), np.array(case["edge_attr"], dtype=np.float64),
                  sd, "c.", case["kind"], case["aggr"])
    np.testing.assert_allclose(out, np.array(case["expected"]), rtol=0, atol=1e-9)


# ------------------------------------------------------------------------------------------------ torch oracle vs numpy oracle
def _random_graph(rng, n, e):
    ei = rng.integers(0, n, size=(2, e))
    ei[1, ei[1] == n - 1] = 0                                   # node n-1 never receives an edge (empty segment)
    ei[:, -1] = ei[:, 0]                                        # one duplicated edge
    return ei


@pytest.mark.parametrize("conv_type,aggr,bn_mlps,enc", [("MPNNConv", "max", False, False), ("MPNNConv", "mean", True, False),
                                                        ("MPNNConv", "add", False, True), ("RadarPointGNNConv", "max", False, False),
                                                        ("RadarPointGNNConv", "mean", True, False)])
def test_det_net_basic_torch_oracle_equals_numpy_oracle(conv_type, aggr, bn_mlps, enc):
    from radargnn_amd import gnn
    rng = np.random.default_rng(5)
    n, e = 40, 160
    dims = [12, 12] if conv_type == "RadarPointGNNConv" else [10, 7]
    cfg = gnn.GNNArchitectureConfig(5, 2, dims, [4], [6, 5], True, True, [8, 12], [3, 6], conv_type, bn_mlps,
                                    aggregation_function=aggr, conv_use_edge_encoder=enc)
    torch.manual_seed(11)
    model = gnn.DetNetBasic(cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():                                     # non-trivial BatchNorm affine parameters
        if k.endswith("module.weight"):
            sd[k] = v + torch.linspace(-0.3, 0.4, v.numel())
        if k.endswith("module.bias"):
            sd[k] = v + torch.linspace(0.2, -0.1, v.numel())
    x = torch.from_numpy(rng.normal(size=(n, 5)))
    ei = torch.from_numpy(_random_graph(rng, n, e))
    ea = torch.from_numpy(rng.normal(size=(e, 2)))
    c_t, b_t = G.det_net_basic(x, ei, ea, sd, conv_type, aggr, dtype=torch.float64)
    c_n, b_n = GN.det_net_basic(x, ei, ea, sd, conv_type, aggr)
    np.testing.assert_allclose(c_t.numpy(), c_n, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(b_t.numpy(), b_n, rtol=1e-10, atol=1e-10)


# ------------------------------------------------------------------------------------------------ the reference itself, if it can run
def test_reference_layers_if_torch_geometric_is_present():
    pytest.importorskip("torch_geometric")
    ref_src = "/root/reference/src"
    if not os.path.isdir(ref_src):
        pytest.skip("reference sources not present on this machine")
    # (the repo's gnnradarobjectdetection/ shim must not shadow the reference package for this one test)
    saved = {k: v for k, v in sys.modules.items() if k.startswith("gnnradarobjectdetection")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, ref_src)
    try:
        from gnnradarobjectdetection.gnn import mpnn_layers as ref
        assert os.path.abspath(ref.__file__).startswith(ref_src)
        rng = np.random.default_rng(0)
        n, e = 30, 120
        x = torch.from_numpy(rng.normal(size=(n, 6))).float()
        ei = torch.from_numpy(_random_graph(rng, n, e))
        ea = torch.from_numpy(rng.normal(size=(e, 3))).float()
        for aggr in ("max", "mean", "add"):
            torch.manual_seed(1)
            layer = ref.MPNNConv(6, 9, 3, aggr=aggr, post_layers=2)
            sd = {"c." + k: v for k, v in layer.state_dict().items()}
            torch.testing.assert_close(layer(x, ei, ea), G.mpnn_conv(x, ei, ea, sd, "c.", aggr), rtol=1e-5, atol=1e-5)
            layer = ref.RadarPointGNNConv(6, 3, aggr=aggr)
            sd = {"c." + k: v for k, v in layer.state_dict().items()}
            torch.testing.assert_close(layer(x, ei, ea), G.radar_point_gnn_conv(x, ei, ea, sd, "c.", aggr), rtol=1e-5, atol=1e-5)
    finally:
        sys.path.remove(ref_src)
        for k in [k for k in sys.modules if k.startswith("gnnradarobjectdetection")]:
            del sys.modules[k]
        sys.modules.update(saved)


@pytest.mark.parametrize("conv,aggr,enc", [("MPNNConv", "max", False), ("MPNNConv", "mean", True), ("MPNNConv", "add", False),
                                           ("RadarPointGNNConv", "max", False), ("RadarPointGNNConv", "add", False)])
def test_hoisted_float64_evaluation_equals_the_faithful_oracle(conv, aggr, enc):
    """oracle/gnn_hoisted.py (what the full-size GPU parity tests compare against) is the same function as the faithful per-edge
    oracle: random graph with isolated targets and duplicate edges, train-mode BatchNorm, chunked edge stage."""
    from oracle import gnn_hoisted as GH
    from radargnn_amd import gnn
    torch.manual_seed(11)
    widths = [24, 24] if conv == "RadarPointGNNConv" else [40, 24, 16]
    cfg = gnn.GNNArchitectureConfig(5, 3, widths, [6], [8, 5], True, True, [16, 24], [4, 6], conv, False, 1, 2, enc, aggr)
    sd = {k: v.detach().clone() for k, v in gnn.DetNetBasic(cfg).state_dict().items()}
    n, e = 90, 400
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 5, generator=g)
    ei = torch.randint(0, n - 10, (2, e), generator=g)             # the last 10 nodes have no edges at all
    ei[:, -7:] = ei[:, :7]                                          # duplicate edges
    ea = torch.randn(e, 3, generator=g)
    c0, b0 = G.det_net_basic(x, ei, ea, sd, conv_layer_type=conv, aggr=aggr, dtype=torch.float64)
    c1, b1 = GH.det_net_basic_hoisted(x, ei, ea, sd, conv_layer_type=conv, aggr=aggr, chunk=64)
    assert (c1 - c0).abs().max().item() <= 1e-11 * c0.abs().max().item()
    assert (b1 - b0).abs().max().item() <= 1e-11 * b0.abs().max().item()
