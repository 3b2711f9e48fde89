"""Hand-derived known answers for the two conv layers with ASYMMETRIC weights (data for tests/test_oracle_gnn.py and
tests/test_gpu_gnn.py).  Derived on paper from gnn/mpnn_layers.py:86-101 (MPNNConv) and :171-184 (RadarPointGNNConv) of the
reference plus torch_geometric's documented "source_to_target" flow (x_j = x[edge_index[0]], x_i = x[edge_index[1]],
reduction per edge_index[1], empty segment -> 0).

Graph (all cases): 3 nodes, x = [1, 2, 3]; 4 edges  source -> target (attribute):
    e0: 0 -> 1 (10)     e1: 2 -> 1 (20)     e2: 1 -> 0 (30)     e3: 0 -> 1 (5)      (e3 repeats the node pair of e0)
node 2 receives nothing.

MPNNConv(1, 1, 1):  message = W_pre [x_i, x_j, e] + b_pre,
    W_pre = [[1, 10, 100], [2, 0, -1], [0, 1, 0.5]],  b_pre = [0.5, -0.5, 0]
    e0: [2, 1, 10] -> [2 + 10 + 1000 + 0.5, 4 - 10 - 0.5, 1 + 5]   = [1012.5,  -6.5,  6  ]
    e1: [2, 3, 20] -> [2 + 30 + 2000 + 0.5, 4 - 20 - 0.5, 3 + 10]  = [2032.5, -16.5, 13  ]
    e3: [2, 1,  5] -> [2 + 10 +  500 + 0.5, 4 -  5 - 0.5, 1 + 2.5] = [ 512.5,  -1.5,  3.5]
    e2: [1, 2, 30] -> [1 + 20 + 3000 + 0.5, 2 - 30 - 0.5, 2 + 15]  = [3021.5, -28.5, 17  ]
  (with the columns swapped to [x_j, x_i, e], e0 would give 1 + 20 + 1000.5 = 1021.5: the vectors tell the orders apart)
  node 1 reduces {e0, e1, e3}:  max [2032.5, -1.5, 13]   mean [3557.5, -24.5, 22.5] / 3   add [3557.5, -24.5, 22.5]
  node 0 reduces {e2}: [3021.5, -28.5, 17];  node 2: [0, 0, 0]
  update = W_post [x, m] + b_post,  W_post = [[1, 0.001, -1, 2]],  b_post = [0.25]
    max:  node 0: 1 + 3.0215 + 28.5 + 34 + 0.25 = 66.7715;  node 1: 2 + 2.0325 + 1.5 + 26 + 0.25 = 31.7825;  node 2: 3.25
    mean: node 1: 2 + 3.5575 / 3 + 24.5 / 3 + 15 + 0.25 = 26.6025 (= 2 + 1.18583.. + 8.16666.. + 15 + 0.25)
    add:  node 1: 2 + 3.5575 + 24.5 + 45 + 0.25 = 75.3075

RadarPointGNNConv(1, 1):  message = W_pre [x_j, e] + b_pre,  W_pre = [[1, 10], [-1, 2]],  b_pre = [0, 1]
    e0: [1, 10] -> [101, 20]   e1: [3, 20] -> [203, 38]   e3: [1, 5] -> [51, 10]   e2: [2, 30] -> [302, 59]
  node 1: max [203, 38], mean [355, 68] / 3;  node 0: [302, 59];  node 2: [0, 0]
  update = W_post [x, m] + b_post + x,  W_post = [[2, 0.01, -0.5]],  b_post = [1]
    max:  node 0: 2 + 3.02 - 29.5 + 1 + 1 = -22.48;  node 1: 4 + 2.03 - 19 + 1 + 2 = -9.97;  node 2: 6 + 1 + 3 = 10
    mean: node 1: 4 + 3.55 / 3 - 34 / 3 + 1 + 2 = -3.15
"""
_GRAPH = {"x": [[1.0], [2.0], [3.0]], "edge_index": [[0, 2, 1, 0], [1, 1, 0, 1]], "edge_attr": [[10.0], [20.0], [30.0], [5.0]]}
_MPNN = {"c.pre_mlp.0.weight": [[1, 10, 100], [2, 0, -1], [0, 1, 0.5]], "c.pre_mlp.0.bias": [0.5, -0.5, 0],
         "c.post_mlp.0.weight": [[1, 0.001, -1, 2]], "c.post_mlp.0.bias": [0.25]}
_RPG = {"c.pre_mlp.0.weight": [[1, 10], [-1, 2]], "c.pre_mlp.0.bias": [0, 1],
        "c.post_mlp.0.weight": [[2, 0.01, -0.5]], "c.post_mlp.0.bias": [1]}


def _case(name, kind, aggr, sd, expected):
    return dict(_GRAPH, name=name, kind=kind, aggr=aggr, state_dict=sd, expected=[[v] for v in expected])


CASES = [
    _case("mpnn-max", "MPNNConv", "max", _MPNN, [66.7715, 31.7825, 3.25]),
    _case("mpnn-mean", "MPNNConv", "mean", _MPNN, [66.7715, 26.6025, 3.25]),
    _case("mpnn-add", "MPNNConv", "add", _MPNN, [66.7715, 75.3075, 3.25]),
    _case("rpg-max", "RadarPointGNNConv", "max", _RPG, [-22.48, -9.97, 10.0]),
    _case("rpg-mean", "RadarPointGNNConv", "mean", _RPG, [-22.48, -3.15, 10.0]),
]
