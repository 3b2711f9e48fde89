"""Second, independent CPU restatement of the GNN half: float64 numpy, one Python iteration per edge.  TEST INFRASTRUCTURE ONLY
(imported by tests/ only; the product never does).

SURVEY.md section 8(c) asks for it because torch_geometric cannot be imported in this image: ``oracle/gnn_oracle.py`` (torch,
vectorised gather / cat / F.linear / scatter_reduce) is then the only statement of the PyG semantics the HIP path is checked
against.  This module shares no code and no library call with it -- explicit loops, Python lists per target, numpy matmuls --
so a mistake would have to be made twice, independently, to go unnoticed (tests/test_oracle_gnn.py compares the two on random
graphs, and both against hand-derived answers).

Semantics restated (torch_geometric 2.1.0.post1, flow "source_to_target"):
* edge e = (edge_index[0, e] -> edge_index[1, e]); ``x_j`` is the SOURCE row, ``x_i`` the TARGET row; messages are reduced per
  target; a target without incoming edges receives the zero vector; "mean" divides by the number of incoming edges;
* ``MPNNConv.message`` (gnn/mpnn_layers.py:94-101): pre_mlp(cat[x_i, x_j, (edge_encoder) e]); ``forward`` (:86-92):
  post_mlp(cat[x, m]);
* ``RadarPointGNNConv.message`` (:179-184): pre_mlp(cat[x_j, e]); ``forward`` (:171-177): post_mlp(cat[x, m]) + x;
* ``get_mlp`` (gnn/gnn_models.py:137-178): Linear, then ([BatchNorm], ReLU, Linear) per further width;
* ``DetNetBasic.forward`` (gnn/gnn_models.py:104-134): embeddings, per layer conv -> BatchNorm (batch statistics, biased
  variance, eps 1e-5) -> ReLU, then the two heads.
"""
from typing import Dict, List

import numpy as np


def _f64(a):
    return np.asarray(a.detach().cpu().numpy() if hasattr(a, "detach") else a, dtype=np.float64)


def _layers(sd: Dict[str, np.ndarray], prefix: str) -> List[tuple]:
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)})
    out, expect = [], 0
    for i in idx:
        out += [("relu",)] * (i - expect)
        base = f"{prefix}{i}."
        if base + "module.weight" in sd:
            out.append(("bn", base + "module."))
        else:
            out.append(("lin", sd[base + "weight"], sd.get(base + "bias")))
        expect = i + 1
    return out


def _bn_train(h: np.ndarray, gamma: np.ndarray, beta: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    mean = h.sum(axis=0) / h.shape[0]
    var = ((h - mean) ** 2).sum(axis=0) / h.shape[0]               # biased, as BatchNorm normalises with
    return (h - mean) / np.sqrt(var + eps) * gamma + beta


def _bn_eval(h, gamma, beta, rm, rv, eps: float = 1e-5):
    return (h - rm) / np.sqrt(rv + eps) * gamma + beta


def mlp(x: np.ndarray, sd, prefix: str, training: bool = True) -> np.ndarray:
    for layer in _layers(sd, prefix):
        if layer[0] == "relu":
            x = np.where(x > 0, x, 0.0)
        elif layer[0] == "lin":
            x = x @ layer[1].T + (0.0 if layer[2] is None else layer[2])
        else:
            p = layer[1]
            x = (_bn_train(x, sd[p + "weight"], sd[p + "bias"]) if training
                 else _bn_eval(x, sd[p + "weight"], sd[p + "bias"], sd[p + "running_mean"], sd[p + "running_var"]))
    return x


def _reduce(rows: List[np.ndarray], width: int, aggr: str) -> np.ndarray:
    if not rows:
        return np.zeros(width)
    stack = np.stack(rows)
    if aggr == "max":
        return stack.max(axis=0)
    if aggr in ("add", "sum"):
        return stack.sum(axis=0)
    if aggr == "mean":
        return stack.sum(axis=0) / len(rows)
    raise ValueError(aggr)


def conv(x: np.ndarray, edge_index: np.ndarray, edge_attr: np.ndarray, sd, prefix: str, kind: str, aggr: str) -> np.ndarray:
    n = x.shape[0]
    inbox: List[List[np.ndarray]] = [[] for _ in range(n)]
    width = None
    for e in range(edge_index.shape[1]):
        s, t = int(edge_index[0, e]), int(edge_index[1, e])
        a = edge_attr[e]
        if kind == "MPNNConv":
            if prefix + "edge_encoder.weight" in sd:
                a = sd[prefix + "edge_encoder.weight"] @ a + sd[prefix + "edge_encoder.bias"]
            m = np.concatenate([x[t], x[s], a])
        else:
            m = np.concatenate([x[s], a])
        m = mlp(m[None, :], sd, prefix + "pre_mlp.")[0]
        width = m.shape[0]
        inbox[t].append(m)
    if width is None:                                              # no edges at all: width from the first pre_mlp weight
        width = sd[prefix + "pre_mlp.0.weight"].shape[0]
    agg = np.stack([_reduce(inbox[t], width, aggr) for t in range(n)]) if n else np.zeros((0, width))
    h = mlp(np.concatenate([x, agg], axis=1), sd, prefix + "post_mlp.")
    return h + x if kind == "RadarPointGNNConv" else h


def det_net_basic(x, edge_index, edge_attr, state_dict, conv_layer_type: str = "MPNNConv", aggr: str = "max",
                  training: bool = True):
    sd = {k: _f64(v) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
    x, ea, ei = _f64(x), _f64(edge_attr), np.asarray(edge_index.detach().cpu().numpy() if hasattr(edge_index, "detach") else edge_index)
    if any(k.startswith("node_emb_mlp.") for k in sd):
        x = mlp(x, sd, "node_emb_mlp.", training)
    if any(k.startswith("edge_emb_mlp.") for k in sd):
        ea = mlp(ea, sd, "edge_emb_mlp.", training)
    for l in range(len({k.split(".")[1] for k in sd if k.startswith("convs.")})):
        h = conv(x, ei, ea, sd, f"convs.{l}.", conv_layer_type, aggr)
        p = f"batch_norms.{l}.module."
        h = (_bn_train(h, sd[p + "weight"], sd[p + "bias"]) if training
             else _bn_eval(h, sd[p + "weight"], sd[p + "bias"], sd[p + "running_mean"], sd[p + "running_var"]))
        x = np.where(h > 0, h, 0.0)
    return mlp(x, sd, "classification_head.", training), mlp(x, sd, "regression_head.", training)
