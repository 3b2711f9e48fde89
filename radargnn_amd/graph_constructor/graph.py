"""``Graph`` / ``GeometricGraph`` with the public surface of the reference's
``graph_constructor/graph.py`` (attributes ``X, V, F, A, E, X_feat, E_feat``; methods ``build``,
``add_node_features``, ``get_degree``, ``add_invariant_feature``, ``add_degree_to_inv_features``,
``extract_node_pair_features``, ``extract_single_node_features``, ``show``) -- numpy arrays in and out, the
work done by librgnn.so on the MI355X:

    build                         grid-hash kNN / radius search      (replaces sklearn KD-tree + toarray, graph.py:52-82)
    get_degree                    undirected-degree kernel            (replaces networkx, graph.py:93-96)
    extract_node_pair_features    one thread per edge                 (replaces the Python loop, graph.py:172-223)
    extract_single_node_features  one thread per node                 (graph.py:225-275)

Differences a caller can observe: ``A`` (the dense N x N adjacency, 72 MB at N = 3000) is built lazily on first
access instead of on every ``build``; within a row, radius neighbours come out index-ascending (the reference's
order there is the KD-tree's traversal order).  There is no CPU fallback.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import ops


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("radargnn_amd: no MI355X visible; the graph constructor runs on the HIP path only "
                           "(no CPU fallback)")
    return torch.device("cuda")


def _f64(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(_device())


def _distance_basis(X: np.ndarray) -> np.ndarray:
    """The reference measures distances over ALL columns of ``X`` (graph.py:45-50,57-58: whatever the caller concatenated).  The HIP
    search is compiled for 2 (X), 4 (X | V) and 8 columns; other widths up to 8 are padded with zero columns to the next of these --
    exact: the KD-tree's reduced distance accumulates t * t per dimension in order, and a zero column adds +0.0.  More than 8 columns
    are refused (the grid bins on the first two columns; no shipped configuration comes near)."""
    w = X.shape[1]
    if w in (2, 4, 8):
        return X
    if 1 <= w < 8:
        to = 2 if w < 2 else (4 if w < 4 else 8)
        return np.concatenate([np.asarray(X, dtype=np.float64), np.zeros((X.shape[0], to - w))], axis=1)
    raise ValueError(f"the HIP neighbour search supports distance bases of 1 to 8 columns (got {w}): the shipped configurations use "
                     "2 (X) or 4 (X | V)")


class Graph:
    """General graph: ``E`` int32 [n_edges, 2] with rows (query i, neighbour j)."""

    def __init__(self):
        self.X_feat = None
        self.E_feat = None
        self.E = None
        self._A = None
        self._n_nodes = None

    # dense adjacency on demand -------------------------------------------------------------------
    @property
    def A(self):
        if self._A is None and self.E is not None:
            n = self._n_nodes if self._n_nodes is not None else int(self.E.max()) + 1
            A = np.zeros((n, n), dtype=np.float64)
            A[self.E[:, 0], self.E[:, 1]] = 1.0
            self._A = A
        return self._A

    @A.setter
    def A(self, value):
        self._A = value

    def build(self, X: np.ndarray, routine: str, k: int = 6, r: float = 1) -> None:
        """Edges between the rows of ``X`` (all columns count for the distance); nothing happens for fewer than
        two points or an unknown routine (graph.py:45-50)."""
        X = np.asarray(X)
        n = X.shape[0]
        if n <= 1 or routine not in ("knn", "radius"):
            return
        Xd = _f64(_distance_basis(X))
        ptr = torch.tensor([0, n], dtype=torch.int64, device=Xd.device)
        if routine == "knn":
            if k >= n:                                          # what sklearn raises under the reference
                raise ValueError(f"Expected n_neighbors < n_samples_fit, but n_neighbors = {k}, "
                                 f"n_samples_fit = {n}, n_samples = {n}")
            _, ei, _ = ops.knn_graph(Xd, ptr, int(k))
        else:
            _, _, ei = ops.radius_graph(Xd, ptr, float(r))
        self.E = np.ascontiguousarray(ei.t().to(torch.int32).cpu().numpy())
        self._A = None
        self._n_nodes = n

    def add_node_features(self, feat: np.ndarray) -> None:
        if self.X_feat is None:
            self.X_feat = feat
        elif feat.shape[0] == self.X_feat.shape[0]:
            self.X_feat = np.concatenate((self.X_feat, feat), axis=1)
        else:
            raise Exception("Feature dimension not compatible")          # graph.py:91

    def get_degree(self) -> list:
        """Undirected degree |{j : i->j or j->i}| per node (what networkx reports for the adjacency)."""
        n = self._n_nodes
        # out-CSR of the directed edges = "CSR by target" of the reversed edge list (stable, any row order of E)
        rev = torch.from_numpy(np.ascontiguousarray(self.E[:, ::-1].T, dtype=np.int64)).to(_device())
        rowptr, col, _ = ops.csr_by_target(rev, n)
        return ops.undirected_degree(rowptr, col, n).cpu().tolist()

    def show(self, node_size: float = 60) -> None:
        import matplotlib.pyplot as plt
        import networkx as nx
        G = nx.from_numpy_array(self.A)
        _, ax = plt.subplots()
        nx.draw(G, ax=ax, node_size=node_size)


class GeometricGraph(Graph):
    """Graph over radar points: spatial coordinates ``X``, velocities ``V``, invariant features ``F`` (dict)."""

    def __init__(self):
        super().__init__()
        self.X = None
        self.V = None
        self.F = None

    def build(self, X, routine, k: int = 6, r: float = 1) -> None:
        super().build(X, routine, k=k, r=r)
        if self._n_nodes is None and self.X is not None:
            self._n_nodes = np.asarray(self.X).shape[0]

    def add_invariant_feature(self, name: str, F_add: np.ndarray) -> None:
        if self.F is None:
            self.F = {name: F_add}
        else:
            self.F[name] = F_add

    def add_degree_to_inv_features(self) -> None:
        deg = self.get_degree()
        self.add_invariant_feature("degree", np.array([deg]).reshape(len(deg), 1))

    def extract_node_pair_features(self, features: List[str], edge_mode: str) -> None:
        """Fill ``E_feat`` (float64 [n_edges, De]) with the requested per-edge features in list order:
        point_pair_features (4), spatial_euclidean_distance, velocity_euclidean_distance, relative_position (2),
        relative_velocity (2)."""
        X, V = _f64(self.X), _f64(self.V)
        ei = torch.from_numpy(np.ascontiguousarray(self.E.T, dtype=np.int64)).to(X.device)
        out, status = ops.edge_features(X, V, ei, features, edge_mode, dtype=torch.float64)   # raises "Invalid feature specified"
        if status.item() & ops.STATUS_DOT_PRODUCT:
            raise Exception("Error in dot product calculation")          # features.py:56,77,91
        res = out.cpu().numpy()
        if self.E_feat is None:
            self.E_feat = res
        else:
            self.E_feat[:, :] = res                                      # an existing buffer is reused (graph.py:168-169)

    def extract_single_node_features(self, features: List[str]) -> None:
        """Append the requested per-node features to ``X_feat`` in list order (rcs, time_index, degree,
        velocity_vector_length, velocity_vector (2), spatial_coordinates (2))."""
        if "degree" in features:
            self.add_degree_to_inv_features()
        names = list(features)
        X = _f64(self.X)
        get = lambda key: None if (self.F is None or self.F.get(key) is None) else _f64(np.asarray(self.F.get(key)).reshape(-1))
        deg = None
        if "degree" in names:
            deg = torch.from_numpy(np.asarray(self.F["degree"]).reshape(-1).astype(np.int32)).to(X.device)
        V = None if self.V is None else _f64(self.V)
        feat = ops.node_features(X, V, get("rcs"), get("time_index"), deg, names, dtype=torch.float64).cpu().numpy()
        if self.X_feat is None:
            self.X_feat = feat
        else:
            self.X_feat = np.concatenate((self.X_feat, feat), axis=1)

    def show(self, node_size: float = 60, show_velocity_vector: bool = False, vec_scale: float = 10,
             with_labels: bool = False) -> None:
        import matplotlib.pyplot as plt
        import networkx as nx
        G = nx.Graph()
        for i, x in enumerate(self.X):
            G.add_node(i, pos=x[0:2])
        G.add_edges_from((int(a), int(b)) for a, b in self.E)
        _, ax = plt.subplots()
        nx.draw(G, nx.get_node_attributes(G, "pos"), ax=ax, node_size=node_size, with_labels=with_labels)
        plt.axis("on")
        ax.tick_params(left=True, bottom=True, labelleft=True, labelbottom=True)
        if show_velocity_vector:
            ax.quiver(self.X[:, 0], self.X[:, 1], self.V[:, 0], self.V[:, 1], scale=vec_scale, color="red")


# ---------------------------------------------------------------------------------------------------
# the caller on the pre-processor side (out of scope as a subsystem, its call site is the boundary)
# ---------------------------------------------------------------------------------------------------
def nearest_neighbor_index(X: np.ndarray) -> np.ndarray:
    """Index of the nearest other point of every row of ``X`` -- the k = 1 use of the neighbour search outside the
    graph builder: the E(n)-invariant box representation takes
    ``X_nn = X[np.where(kneighbors_graph(X, 1, include_self=False).toarray() == 1)[1]]`` at
    preprocessor/radarscenes/dataset_creation.py:316-318,532, preprocessor/nuscenes/conversion.py:133-137 and
    postprocessor/postprocessing.py:233-237,469.  Same kernel as ``Graph.build(X, "knn", k=1)`` (f64 distances, ties by
    index); no dense N x N matrix.  Raises what sklearn raises for fewer than two points."""
    X = np.asarray(X)
    n = X.shape[0]
    if n <= 1:
        raise ValueError(f"Expected n_neighbors < n_samples_fit, but n_neighbors = 1, n_samples_fit = {n}, n_samples = {n}")
    Xd = _f64(_distance_basis(X))
    ptr = torch.tensor([0, n], dtype=torch.int64, device=Xd.device)
    nbr, _, _ = ops.knn_graph(Xd, ptr, 1, want_edge_index=False)
    return nbr.view(-1).to(torch.int64).cpu().numpy()


def nearest_neighbor_points(X: np.ndarray) -> np.ndarray:
    """``X_nn`` of the call sites listed at ``nearest_neighbor_index``: row i = coordinates of the nearest neighbour of i."""
    return np.asarray(X)[nearest_neighbor_index(X)]


def time_index_of(timestamp: np.ndarray) -> np.ndarray:
    """Rank of every timestamp among the frame's distinct timestamps, shaped like ``timestamp``
    (radarscenes/dataset_creation.py:214-223)."""
    ts = _f64(np.asarray(timestamp).reshape(-1))
    ptr = torch.tensor([0, ts.numel()], dtype=torch.int64, device=ts.device)
    ti, status = ops.time_index(ts, ptr)
    if status.item() & ops.STATUS_TIME_INDEX_OVERFLOW:
        raise RuntimeError("more than 3072 distinct timestamps in one frame")
    return ti.cpu().numpy().reshape(np.asarray(timestamp).shape)


def build_geometric_graph(config, point_cloud) -> GeometricGraph:
    """Counterpart of ``GraphConstructor.build_geometric_graph`` (radarscenes/dataset_creation.py:190-229;
    nuScenes twin nuscenes/conversion.py:70-109).  ``point_cloud`` needs ``X_cc``, ``V_cc_compensated``, ``rcs``,
    ``timestamp``; ``config`` is a ``GraphConstructionConfiguration``."""
    if config.distance_definition == "X":
        basis = point_cloud.X_cc
    elif config.distance_definition == "XV":
        basis = np.concatenate((point_cloud.X_cc, point_cloud.V_cc_compensated), axis=1)
    else:
        raise ValueError(config.distance_definition)
    graph = GeometricGraph()
    graph.X = point_cloud.X_cc
    graph.V = point_cloud.V_cc_compensated
    graph.F = {"rcs": getattr(point_cloud, "rcs", None)}
    if "time_index" in config.node_features:
        graph.add_invariant_feature("time_index", time_index_of(point_cloud.timestamp))
    graph.build(basis, config.graph_construction_algorithm, k=config.k, r=config.r)
    graph.extract_node_pair_features(config.edge_features, config.edge_mode)
    graph.extract_single_node_features(config.node_features)
    return graph


def create_graph_tensors(graph: GeometricGraph, point_cloud=None) -> dict:
    """dtype / layout hand-off of ``create_graph_data`` (radarscenes/dataset_creation.py:786-814) without the
    torch_geometric ``Data`` container: x f32 [N,Dn], edge_index int64 [2,E], edge_attr f32 [E,De], pos, vel."""
    out = {
        "x": torch.tensor(graph.X_feat, dtype=torch.float32),
        "edge_index": torch.tensor(graph.E.T, dtype=torch.long),
        "edge_attr": torch.tensor(graph.E_feat, dtype=torch.float32),
    }
    if point_cloud is not None:
        out["pos"] = torch.tensor(point_cloud.X_cc, dtype=torch.float32)
        out["vel"] = torch.tensor(point_cloud.V_cc_compensated, dtype=torch.float32)
    return out
