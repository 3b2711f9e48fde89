"""Reference-SHAPED CPU path for timing.  TEST / BENCH INFRASTRUCTURE ONLY (bench.py ``cpu_baseline`` leg).

Same library calls and loop structure as the reference, so its wall time is what a user of the reference pays on
the host cores of the same box:

* graph:  scikit-learn ``kneighbors_graph`` / ``radius_neighbors_graph`` -> dense ``toarray()`` -> ``nonzero()``
          (graph_constructor/graph.py:52-82), networkx degree from the dense adjacency (graph.py:93-96), ONE PYTHON
          ITERATION PER EDGE for the edge features (graph.py:172-223) calling the point-pair routine per edge
          (features.py:6-122), column concatenation for the node features (graph.py:225-275);
* model:  eager PyTorch gather -> cat -> Linear -> scatter per layer (oracle/gnn_oracle.py, the faithful per-edge
          form of gnn/mpnn_layers.py), float32, torch intra-op threads = all cores.

It is a restatement written for this repo (the reference's sources never travel to the GPU box); its outputs
are checked against oracle/graph_oracle.py in tests/test_reference_shaped.py.
"""
from __future__ import annotations

import time
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import gnn_oracle, graph_oracle


def _edge_list_sklearn(X: np.ndarray, algorithm: str, k: int, r: float):
    from sklearn.neighbors import kneighbors_graph, radius_neighbors_graph
    if algorithm == "knn":
        sp = kneighbors_graph(X, k, mode="connectivity", include_self=False)
    else:
        sp = radius_neighbors_graph(X, r, mode="connectivity", include_self=False)
    dense = sp.toarray()                                   # the reference materialises the N x N matrix
    rows = sp.nonzero()[0].reshape(-1, 1)
    cols = sp.nonzero()[1].reshape(-1, 1)
    return np.concatenate((rows, cols), axis=1), dense


def _degree_networkx(dense: np.ndarray) -> np.ndarray:
    import networkx as nx
    g = nx.from_numpy_array(dense)
    return np.array([d for _, d in g.degree()]).reshape(-1, 1)


def _pair_row(xi, xj, vi, vj, names: Sequence[str], mode: str) -> List[float]:
    row: List[float] = []
    for name in names:
        if name == "point_pair_features":
            vals = graph_oracle.point_pair_features(xi.reshape(1, 2), xj.reshape(1, 2), vi.reshape(1, 2),
                                                    vj.reshape(1, 2), mode)
            row.extend(float(v[0]) for v in vals)
        elif name == "spatial_euclidean_distance":
            row.append(float(np.linalg.norm(xi - xj, ord=2)))
        elif name == "velocity_euclidean_distance":
            row.append(float(np.linalg.norm(vi - vj, ord=2)))
        elif name in ("relative_position", "relative_velocity"):
            a, b = (xi, xj) if name == "relative_position" else (vi, vj)
            d0, d1 = a[0] - b[0], a[1] - b[1]
            row.extend([d0, d1] if mode == "directed" else [abs(d0), abs(d1)])
        else:
            raise Exception("Invalid feature specified")
    return row


def build_frame_graph(X, V, rcs, timestamp, algorithm, k, r, node_names, edge_names, mode) -> Dict[str, np.ndarray]:
    E, dense = _edge_list_sklearn(X, algorithm, k, r)
    width = sum(graph_oracle.EDGE_FEATURE_WIDTH[n] for n in edge_names)
    E_feat = np.empty((E.shape[0], width))
    for idx, (i, j) in enumerate(E):                       # one Python iteration per edge, like graph.py:172
        E_feat[idx, :] = _pair_row(X[int(i)], X[int(j)], V[int(i)], V[int(j)], edge_names, mode)
    F = {"rcs": rcs}
    if "time_index" in node_names:
        stamps = np.unique(timestamp)
        t_idx = np.zeros_like(timestamp)
        for rank, s in enumerate(stamps):
            t_idx[np.where(timestamp == s)[0]] = rank
        F["time_index"] = t_idx
    deg = _degree_networkx(dense) if "degree" in node_names else None
    X_feat = graph_oracle.node_features(X, V, F, deg, node_names)
    return {"E": E, "x": X_feat.astype(np.float32), "edge_index": np.ascontiguousarray(E.T.astype(np.int64)),
            "edge_attr": E_feat.astype(np.float32)}


def time_hot_path(frames, algorithm, k, r, node_names, edge_names, mode, state_dict, conv_layer_type="MPNNConv",
                  aggr="max", repeats: int = 1) -> Dict[str, float]:
    """Wall time of graph-build + forward for ``frames`` as ONE batch.  Returns seconds per stage."""
    t0 = time.perf_counter()
    for _ in range(repeats):
        graphs = [build_frame_graph(f.X, f.V, f.rcs, f.timestamp, algorithm, k, r, node_names, edge_names, mode)
                  for f in frames]
        batch = graph_oracle.collate(graphs)
    t1 = time.perf_counter()
    x = torch.from_numpy(batch["x"])
    ei = torch.from_numpy(batch["edge_index"])
    ea = torch.from_numpy(batch["edge_attr"])
    sd = {k_: v.detach().cpu() for k_, v in state_dict.items()}
    with torch.no_grad():
        gnn_oracle.det_net_basic(x, ei, ea, sd, conv_layer_type, aggr)          # warm-up (thread pool, allocator)
        t2 = time.perf_counter()
        for _ in range(repeats):
            gnn_oracle.det_net_basic(x, ei, ea, sd, conv_layer_type, aggr)
        t3 = time.perf_counter()
    return {"graph_s": (t1 - t0) / repeats, "forward_s": (t3 - t2) / repeats, "frames": len(frames)}


def time_forward_only(frames, algorithm, k, r, node_names, edge_names, mode, state_dict, conv_layer_type="MPNNConv",
                      aggr="max", repeats: int = 1) -> float:
    """Forward time alone (graphs from the vectorised oracle), for trying several torch thread counts."""
    graphs = [graph_oracle.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, algorithm, k, r, node_names, edge_names, mode)
              for f in frames]
    batch = graph_oracle.collate(graphs)
    x, ei, ea = (torch.from_numpy(batch[k_]) for k_ in ("x", "edge_index", "edge_attr"))
    sd = {k_: v.detach().cpu() for k_, v in state_dict.items()}
    with torch.no_grad():
        gnn_oracle.det_net_basic(x, ei, ea, sd, conv_layer_type, aggr)
        t0 = time.perf_counter()
        for _ in range(repeats):
            gnn_oracle.det_net_basic(x, ei, ea, sd, conv_layer_type, aggr)
    return (time.perf_counter() - t0) / repeats


def time_vectorised(frames, algorithm, k, r, node_names, edge_names, mode, state_dict, conv_layer_type="MPNNConv",
                    aggr="max") -> Dict[str, float]:
    """Honesty check: the numpy-vectorised oracle (no dense adjacency, no Python edge loop) + the same forward."""
    t0 = time.perf_counter()
    graphs = [graph_oracle.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, algorithm, k, r, node_names, edge_names, mode)
              for f in frames]
    batch = graph_oracle.collate(graphs)
    t1 = time.perf_counter()
    sd = {k_: v.detach().cpu() for k_, v in state_dict.items()}
    with torch.no_grad():
        gnn_oracle.det_net_basic(torch.from_numpy(batch["x"]), torch.from_numpy(batch["edge_index"]),
                                 torch.from_numpy(batch["edge_attr"]), sd, conv_layer_type, aggr)
    t2 = time.perf_counter()
    return {"graph_s": t1 - t0, "forward_s": t2 - t1, "frames": len(frames)}
