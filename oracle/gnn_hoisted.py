"""Hoisted float64 evaluation of ``DetNetBasic`` for FULL-SIZE parity checks.  TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module; the product (``radargnn_amd``) never does.

``oracle/gnn_oracle.py`` restates the reference's layers in their faithful per-edge form -- gather both endpoint rows,
concatenate, run the message Linear on every edge ([E, D] x [D, D]) -- which at the sizes ``bench.py`` times (C2: E = 799 078,
D = 464; C3: E = 3 072 000) is 170 - 660 GFLOP of float64 per layer plus E x D x 8 B per intermediate: minutes to hours.  This
file evaluates the SAME function through the algebraic identity of SURVEY.md section 0.5,

    aggr_e (W_i x_t + W_j x_s + W_e a_e + b)  =  1[deg_t > 0] (W_i x_t + b)  +  aggr_e (W_j x_s + W_e a_e)      (max, mean)
                                              =  deg_t (W_i x_t + b)  +  sum_e (W_j x_s + W_e a_e)               (add)

(reference: gnn/mpnn_layers.py:86-101 with ``pre_layers == 1``: the message MLP is one Linear and there is no non-linearity in
front of the aggregation), i.e. node-wise products P = x W_i^T + b, Q = x W_j^T and an edge stage that is evaluated in chunks of
edges -- in float64, on whatever torch device the caller names (the GPU box's host cores, or its GPU through torch's own float64
kernels: in either case NOT through librgnn).

It is trusted only as far as it is pinned: every test that uses it at full size first checks it against the faithful
``gnn_oracle.det_net_basic`` (float64) on a small batch of the same configuration IN THE SAME TEST (agreement to ~1e-12: the two
differ only in float64 summation order).  ``RadarPointGNNConv`` hoists the same way (message = Linear(cat[x_j, e])); deeper
message MLPs (``pre_layers > 1``) cannot be hoisted and raise.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import gnn_oracle as G


def _segment_reduce(rows: torch.Tensor, dst: torch.Tensor, out: torch.Tensor, aggr: str) -> None:
    """Fold a chunk of message rows into ``out`` [N, D] (running result; for max ``out`` starts at -inf)."""
    idx = dst.view(-1, 1).expand(-1, rows.shape[1])
    if aggr == "max":
        out.scatter_reduce_(0, idx, rows, reduce="amax", include_self=True)
    else:
        out.scatter_add_(0, idx, rows)


def _edge_stage(Q: torch.Tensor, We: torch.Tensor, ea: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, n: int, aggr: str,
                chunk: int) -> torch.Tensor:
    """aggr_e (Q[src_e] + W_e a_e) per target, without ever holding [E, D]: chunks of ``chunk`` edges."""
    d = Q.shape[1]
    out = torch.full((n, d), float("-inf"), dtype=Q.dtype, device=Q.device) if aggr == "max" else \
        torch.zeros((n, d), dtype=Q.dtype, device=Q.device)
    for a in range(0, src.numel(), chunk):
        s, t = src[a:a + chunk], dst[a:a + chunk]
        _segment_reduce(Q[s] + ea[a:a + chunk] @ We.t(), t, out, aggr)
    return out


def _conv(x, src, dst, deg, ea, sd, prefix: str, conv_layer_type: str, aggr: str, chunk: int) -> torch.Tensor:
    pre = G._sequential(prefix + "pre_mlp.", sd)
    if len(pre) != 1:
        raise NotImplementedError("the hoisted evaluation needs a single-Linear message MLP (pre_layers == 1)")
    W, b = pre[0][1]["w"], pre[0][1]["b"]
    n, c = x.shape
    has = (deg > 0).to(x.dtype).view(-1, 1)
    if conv_layer_type == "MPNNConv":
        Wi, Wj, We = W[:, :c], W[:, c:2 * c], W[:, 2 * c:]
        if prefix + "edge_encoder.weight" in sd:                 # mpnn_layers.py:96-97: e -> Linear(e), then the same algebra
            ea = F.linear(ea, sd[prefix + "edge_encoder.weight"], sd[prefix + "edge_encoder.bias"])
        P = F.linear(x, Wi, b)
    else:                                                        # RadarPointGNNConv: message = Linear(cat[x_j, e])  :181-182
        Wj, We = W[:, :c], W[:, c:]
        P = (b if b is not None else torch.zeros(W.shape[0], dtype=x.dtype, device=x.device)).view(1, -1).expand(n, -1)
    Q = x @ Wj.t()
    agg = _edge_stage(Q, We, ea, src, dst, n, aggr, chunk)
    if aggr == "max":
        m = torch.where(has.bool(), agg + P, torch.zeros_like(agg))           # empty segment -> 0 (torch-scatter)
    elif aggr == "mean":
        m = agg / deg.clamp(min=1).to(x.dtype).view(-1, 1) + has * P
    else:
        m = agg + deg.to(x.dtype).view(-1, 1) * P
    h = G.run_sequential(torch.cat([x, m], dim=-1), sd, prefix + "post_mlp.")
    return h + x if conv_layer_type != "MPNNConv" else h


def det_net_basic_hoisted(x, edge_index, edge_attr, sd: Dict[str, torch.Tensor], conv_layer_type: str = "MPNNConv",
                          aggr: str = "max", training: bool = True, device="cpu", chunk: int = 1 << 18):
    """``DetNetBasic.forward`` (gnn_models.py:104-134) in float64 through the hoisted conv layers.  Same arguments and
    state_dict convention as ``gnn_oracle.det_net_basic``; returns CPU float64 tensors."""
    dev = torch.device(device)
    f64 = torch.float64
    sd = {k: (v.to(dev, f64) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
    x = x.to(dev, f64)
    ea = edge_attr.to(dev, f64)
    ei = edge_index.to(dev)
    src, dst = ei[0], ei[1]
    deg = torch.bincount(dst, minlength=x.shape[0])
    if any(k.startswith("node_emb_mlp.") for k in sd):
        x = G.run_sequential(x, sd, "node_emb_mlp.", training)
    if any(k.startswith("edge_emb_mlp.") for k in sd):
        ea = G.run_sequential(ea, sd, "edge_emb_mlp.", training)
    n_layers = len({k.split(".")[1] for k in sd if k.startswith("convs.")})
    for l in range(n_layers):
        x = _conv(x, src, dst, deg, ea, sd, f"convs.{l}.", conv_layer_type, aggr, chunk)
        x = torch.relu(G.batch_norm(x, sd, f"batch_norms.{l}.module.", training, update=False))
    c = G.run_sequential(x, sd, "classification_head.", training)
    bb = G.run_sequential(x, sd, "regression_head.", training)
    return c.cpu(), bb.cpu()
