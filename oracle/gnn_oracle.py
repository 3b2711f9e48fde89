"""CPU oracle for the GNN half of the hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``radargnn_amd``) never does.

A plain-PyTorch (CPU, eager, float32 or float64) restatement of the *faithful* per-edge form of the
reference's layers -- gather both endpoint rows, concatenate with the edge attributes, run the
message MLP on every edge, scatter-reduce, concatenate with the node row, run the update MLP:

* ``MPNNConv``           src/gnnradarobjectdetection/gnn/mpnn_layers.py:11-101
* ``RadarPointGNNConv``  src/gnnradarobjectdetection/gnn/mpnn_layers.py:104-184
* ``get_mlp``            src/gnnradarobjectdetection/gnn/gnn_models.py:137-178
* ``DetNetBasic``        src/gnnradarobjectdetection/gnn/gnn_models.py:15-134

The reference inherits its mechanics from torch_geometric 2.1.0.post1 / torch-scatter 2.0.9
(``Dockerfile:21-26``), which are NOT installable in this image, so their semantics are restated
from their published behaviour:

* ``MessagePassing`` with the default ``flow="source_to_target"``: ``x_j = x[edge_index[0]]``,
  ``x_i = x[edge_index[1]]``, aggregation over ``edge_index[1]`` with ``dim_size = N``;
* aggregation "max" | "mean" | "add" ("sum"); an empty segment yields 0; mean divides by max(count, 1);
* ``nn.dense.linear.Linear`` = ``F.linear(x, weight[out,in], bias)``;
* ``BatchNorm`` wraps ``torch.nn.BatchNorm1d(eps=1e-5, momentum=0.1)`` as ``.module``; the reference
  never calls ``.eval()``, so batch statistics are used unless the caller flips ``training``.

Parity pinning: "pinned by construction + the reference's known-answer tests" -- test/test_gnn.py
:9-25 (get_mlp -> [10,10,10]), :79-116 (7 / 72), :119-172 (436 with a duplicate edge, max aggregation)
and :175-221 (edge encoder -> 23) are transcribed in tests/test_oracle_gnn.py.  PyG-executed golden
vectors are unavailable (torch_geometric cannot be imported here); a float64 run of this same
restatement is the cross-check for rounding.

The oracle consumes a ``state_dict`` with the reference's key names, so the same
weights drive the oracle and the HIP modules.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


def scatter_reduce_rows(msg: torch.Tensor, index: torch.Tensor, n: int, aggr: str) -> torch.Tensor:
    """torch-scatter semantics: rows of ``msg`` [E,D] reduced into ``out`` [n,D] keyed by ``index``;
    segments with no entry are 0."""
    d = msg.shape[1]
    out = torch.zeros((n, d), dtype=msg.dtype)
    if msg.shape[0] == 0:
        return out
    idx = index.view(-1, 1).expand(-1, d)
    if aggr == "max":
        out = out.scatter_reduce(0, idx, msg, reduce="amax", include_self=False)
        return out                                   # untouched rows keep the initial 0
    if aggr in ("add", "sum"):
        return out.scatter_add(0, idx, msg)
    if aggr == "mean":
        s = out.scatter_add(0, idx, msg)
        cnt = torch.bincount(index, minlength=n).clamp(min=1).to(msg.dtype)
        return s / cnt.view(-1, 1)
    raise ValueError(aggr)


def _sequential(prefix: str, sd: Dict[str, torch.Tensor]) -> List[Tuple[str, dict]]:
    """Recover the layer list of an ``nn.Sequential`` of Linear / BatchNorm / ReLU from its
    state_dict keys: index gaps are ReLUs (parameter-free)."""
    idxs = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)})
    layers = []
    last = -1
    for i in idxs:
        base = f"{prefix}{i}."
        for _ in range(last + 1, i):
            layers.append(("relu", {}))
        if base + "module.running_mean" in sd:
            layers.append(("bn", {"prefix": base + "module."}))
        else:
            layers.append(("linear", {"w": sd[base + "weight"], "b": sd.get(base + "bias")}))
        last = i
    return layers


def batch_norm(x, sd, prefix, training: bool, momentum: float = 0.1, eps: float = 1e-5, update: bool = True):
    """torch.nn.BatchNorm1d forward: batch mean / biased variance when ``training``; running stats are
    updated in place with the unbiased variance."""
    rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    if not update:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm.to(x.dtype), rv.to(x.dtype), sd[prefix + "weight"].to(x.dtype),
                     sd[prefix + "bias"].to(x.dtype), training=training, momentum=momentum, eps=eps)
    return y


def run_sequential(x, sd, prefix, training=True):
    for kind, p in _sequential(prefix, sd):
        if kind == "relu":
            x = torch.relu(x)
        elif kind == "linear":
            x = F.linear(x, p["w"].to(x.dtype), None if p["b"] is None else p["b"].to(x.dtype))
        else:
            x = batch_norm(x, sd, p["prefix"], training, update=False)
    return x


def mpnn_conv(x, edge_index, edge_attr, sd, prefix: str, aggr: str = "max") -> torch.Tensor:
    """``MPNNConv.forward`` (mpnn_layers.py:86-101)."""
    src, dst = edge_index[0], edge_index[1]
    x_j, x_i = x[src], x[dst]                                     # PyG source_to_target
    e = edge_attr
    if prefix + "edge_encoder.weight" in sd:                       # mpnn_layers.py:96-97
        e = F.linear(e, sd[prefix + "edge_encoder.weight"].to(x.dtype), sd[prefix + "edge_encoder.bias"].to(x.dtype))
    m = torch.cat([x_i, x_j, e], dim=-1)                           # mpnn_layers.py:98
    m = run_sequential(m, sd, prefix + "pre_mlp.")                 # mpnn_layers.py:99
    agg = scatter_reduce_rows(m, dst, x.shape[0], aggr)
    out = torch.cat([x, agg], dim=-1)                              # mpnn_layers.py:89
    return run_sequential(out, sd, prefix + "post_mlp.")           # mpnn_layers.py:90


def radar_point_gnn_conv(x, edge_index, edge_attr, sd, prefix: str, aggr: str = "max") -> torch.Tensor:
    """``RadarPointGNNConv.forward`` (mpnn_layers.py:171-184)."""
    src, dst = edge_index[0], edge_index[1]
    m = torch.cat([x[src], edge_attr], dim=-1)                     # mpnn_layers.py:181
    m = run_sequential(m, sd, prefix + "pre_mlp.")
    agg = scatter_reduce_rows(m, dst, x.shape[0], aggr)
    h = run_sequential(torch.cat([x, agg], dim=-1), sd, prefix + "post_mlp.")
    return h + x                                                   # mpnn_layers.py:177


def det_net_basic(x, edge_index, edge_attr, sd: Dict[str, torch.Tensor], conv_layer_type: str = "MPNNConv",
                  aggr: str = "max", training: bool = True, dtype=torch.float32, return_hidden: bool = False):
    """``DetNetBasic.forward`` (gnn_models.py:104-134) driven by a reference-keyed ``state_dict``.
    Running statistics in ``sd`` are NOT modified (use ``bn_running_update`` for that contract)."""
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    x = x.to(dtype)
    edge_attr = edge_attr.to(dtype)
    if any(k.startswith("node_emb_mlp.") for k in sd):             # gnn_models.py:117-118
        x = run_sequential(x, sd, "node_emb_mlp.", training)
    if any(k.startswith("edge_emb_mlp.") for k in sd):             # gnn_models.py:120-121
        edge_attr = run_sequential(edge_attr, sd, "edge_emb_mlp.", training)
    n_layers = len({k.split(".")[1] for k in sd if k.startswith("convs.")})
    hidden = []
    for l in range(n_layers):                                      # gnn_models.py:124-128
        if conv_layer_type == "MPNNConv":
            x = mpnn_conv(x, edge_index, edge_attr, sd, f"convs.{l}.", aggr)
        else:
            x = radar_point_gnn_conv(x, edge_index, edge_attr, sd, f"convs.{l}.", aggr)
        hidden.append(x)
        x = batch_norm(x, sd, f"batch_norms.{l}.module.", training, update=False)
        x = torch.relu(x)
    c = run_sequential(x, sd, "classification_head.", training)    # gnn_models.py:131
    bb = run_sequential(x, sd, "regression_head.", training)       # gnn_models.py:132
    if return_hidden:
        return c, bb, hidden
    return c, bb


def bn_running_update(h: torch.Tensor, running_mean, running_var, momentum: float = 0.1):
    """What one train-mode BatchNorm1d forward does to its running statistics."""
    n = h.shape[0]
    mean = h.mean(0)
    var_unbiased = h.var(0, unbiased=True) if n > 1 else torch.zeros_like(mean)
    return (1 - momentum) * running_mean + momentum * mean, (1 - momentum) * running_var + momentum * var_unbiased
