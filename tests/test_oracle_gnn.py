"""Known-answer tests of the reference's test/test_gnn.py transcribed onto the CPU oracle
(oracle/gnn_oracle.py).  CPU only.  These five answers are the only GNN results the reference pins."""
import torch

from oracle import gnn_oracle as G


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
