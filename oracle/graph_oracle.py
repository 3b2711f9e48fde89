"""CPU oracle for the graph-construction half of the hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``radargnn_amd``) never does.

This is a numpy (float64) restatement of what the reference computes in
``src/gnnradarobjectdetection/graph_constructor/{graph.py,features.py}`` and of the 25 lines of
``preprocessor/radarscenes/dataset_creation.py:190-229`` that call it.  Every function cites the
reference lines it follows.  It is a *vectorised* restatement: no dense NxN adjacency, no per-edge
Python loop (``oracle/reference_shaped.py`` keeps the reference's loop structure for timing).

Third-party arithmetic under the reference: scikit-learn ``kneighbors_graph`` /
``radius_neighbors_graph`` (KD-tree, float64, reduced distance sum((x-y)^2) accumulated in dimension
order, radius test inclusive ``d2 <= r*r``) and networkx ``Graph.degree``.  Neither is pinned by the
reference; scikit-learn 1.7.2 / networkx 3.4.2 are what the golden vectors in ``tests/golden`` were
generated with (``tests/golden/make_golden.py``).

Parity pinning: every function here is checked against golden vectors produced by importing the
reference's own ``graph_constructor`` package (tests/test_oracle_golden.py) and against the
known answers of the reference's ``test/test_graph_constructor.py`` and
``test/test_preprocessor.py:207-257``.

Order conventions: kNN rows come out (distance asc, index asc) which equals the
reference order whenever the k nearest distances of a row are distinct; radius rows come out index
ascending (the reference's KD-tree traversal order is an artefact, tests compare canonically sorted
edge lists).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

EDGE_FEATURE_WIDTH = {
    "point_pair_features": 4,
    "spatial_euclidean_distance": 1,
    "velocity_euclidean_distance": 1,
    "relative_position": 2,
    "relative_velocity": 2,
}
NODE_FEATURE_WIDTH = {
    "rcs": 1, "time_index": 1, "degree": 1, "velocity_vector_length": 1,
    "velocity_vector": 2, "spatial_coordinates": 2,
}


# --------------------------------------------------------------------------------------------
# neighbour search (graph.py:52-82 -> sklearn KDTree64)
# --------------------------------------------------------------------------------------------
def _reduced_distances(Xq: np.ndarray, X: np.ndarray) -> np.ndarray:
    """sum_j (xq_j - x_j)^2 accumulated in dimension order with separate multiply and add, as the
    KD-tree's ``rdist`` does (float64, no FMA contraction in the x86-64 baseline wheels)."""
    d2 = np.zeros((Xq.shape[0], X.shape[0]), dtype=np.float64)
    for j in range(X.shape[1]):
        t = Xq[:, j:j + 1] - X[None, :, j]
        d2 = d2 + t * t
    return d2


def knn_neighbours(X: np.ndarray, k: int, chunk: int = 1024) -> np.ndarray:
    """k nearest neighbours of every row of X among the other rows -> int32 [N,k].

    Follows ``Graph.__build_knn`` (graph.py:52-66): ``kneighbors_graph(X, k, include_self=False)``;
    sklearn raises when ``k >= N`` (the reference does not guard it)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    if k >= n:
        raise ValueError(f"Expected n_neighbors < n_samples_fit, but n_neighbors = {k}, n_samples_fit = {n}")
    out = np.empty((n, k), dtype=np.int32)
    idx = np.arange(n)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        d2 = _reduced_distances(X[s:e], X)
        d2[np.arange(e - s), idx[s:e]] = np.inf                       # include_self=False
        # (distance asc, index asc): stable argsort on distance with the index order as tiebreak
        order = np.argsort(d2, axis=1, kind="stable")[:, :k]
        out[s:e] = order.astype(np.int32)
    return out


def knn_edges(X: np.ndarray, k: int) -> np.ndarray:
    """``E`` of graph.py:61-63: int32 [N*k,2], rows (i, j), i ascending, j by ascending distance."""
    nbr = knn_neighbours(X, k)
    n = X.shape[0]
    src = np.repeat(np.arange(n, dtype=np.int32), k)
    return np.stack([src, nbr.reshape(-1)], axis=1)


def radius_edges(X: np.ndarray, r: float, chunk: int = 1024) -> np.ndarray:
    """``E`` of graph.py:73-79 (``radius_neighbors_graph(X, r, include_self=False)``): inclusive
    ``d2 <= r*r``, int32 [E,2], rows ascending in (i, j)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = X.shape[0]
    r2 = float(r) * float(r)
    rows, cols = [], []
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        d2 = _reduced_distances(X[s:e], X)
        hit = d2 <= r2
        hit[np.arange(e - s), np.arange(s, e)] = False
        ii, jj = np.nonzero(hit)
        rows.append((ii + s).astype(np.int32))
        cols.append(jj.astype(np.int32))
    if not rows:
        return np.zeros((0, 2), dtype=np.int32)
    return np.stack([np.concatenate(rows), np.concatenate(cols)], axis=1)


def build_edges(X: np.ndarray, routine: str, k: int = 6, r: float = 1) -> Optional[np.ndarray]:
    """Dispatch of ``Graph.build`` (graph.py:32-50): nothing happens for N <= 1 or an unknown routine."""
    if X.shape[0] > 1:
        if routine == "knn":
            return knn_edges(X, k)
        if routine == "radius":
            return radius_edges(X, r)
    return None


def canonical_edges(E: np.ndarray) -> np.ndarray:
    """(row, col)-sorted copy of an edge list; the order two implementations are compared in."""
    if E.shape[0] == 0:
        return E.copy()
    order = np.lexsort((E[:, 1], E[:, 0]))
    return E[order]


def undirected_degree(E: np.ndarray, n: int) -> np.ndarray:
    """networkx degree of ``from_numpy_array(A)`` (graph.py:93-96): the number of distinct j with
    A[i,j] != 0 or A[j,i] != 0 (no self loops on this path)."""
    if E is None or E.shape[0] == 0:
        return np.zeros(n, dtype=np.int64)
    a = E[:, 0].astype(np.int64)
    b = E[:, 1].astype(np.int64)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    und = np.unique(lo * n + hi)
    lo, hi = und // n, und % n
    return np.bincount(lo, minlength=n) + np.bincount(hi, minlength=n)


# --------------------------------------------------------------------------------------------
# point-pair features (features.py:6-122)
# --------------------------------------------------------------------------------------------
class DotProductError(Exception):
    pass


def _unit_or_zero(v: np.ndarray) -> np.ndarray:
    """features.py:24-40 / :62-65: an all-zero vector stays zero, anything else is divided by its
    2-norm (np.linalg.norm(ord=2) = sqrt(sum of squares))."""
    nrm = np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1])
    zero = (v[:, 0] == 0.0) & (v[:, 1] == 0.0)
    safe = np.where(zero, 1.0, nrm)
    out = v / safe[:, None]
    out[zero] = 0.0
    return out


def _dot2(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]


def _clamped_angle_deg(dot: np.ndarray) -> np.ndarray:
    """features.py:49-58: |dot| in (1, 1+1e-3) is clamped to +-1, beyond that the reference raises."""
    over = np.abs(dot) > 1.0
    if np.any(over & ~((np.abs(dot) - 1.0) < 1e-3)):
        raise DotProductError("Error in dot product calculation")
    dot = np.where(over, np.sign(dot), dot)
    return np.arccos(dot) * 180 / np.pi


def point_pair_features(p1, p2, v1, v2, mode: str):
    """Vectorised ``get_En_equivariant_point_pair_metrics`` (features.py:6-122); all inputs [E,2].
    Returns d, theta_v1_v2, theta_d_v_min, theta_d_v_max (degrees)."""
    v1n = _unit_or_zero(v1)
    v2n = _unit_or_zero(v2)
    diff = p1 - p2
    d = np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1])            # features.py:43
    th_v = _clamped_angle_deg(_dot2(v1n, v2n))                               # features.py:46-58
    if mode == "directed":
        dvec = _unit_or_zero(p2 - p1)                                         # features.py:62-65
        th1 = _clamped_angle_deg(_dot2(v1n, dvec))                            # features.py:67-79
        th2 = _clamped_angle_deg(_dot2(v2n, dvec))                            # features.py:81-93
        return d, th_v, th1, th2
    if mode == "undirected":
        # features.py:98-120: no clamp in this branch (NaN possible, like the reference)
        with np.errstate(invalid="ignore"):
            d1 = _unit_or_zero(p1 - p2)
            d2 = _unit_or_zero(p2 - p1)
            a11 = np.arccos(_dot2(v1n, d1)) * 180 / np.pi
            a12 = np.arccos(_dot2(v2n, d1)) * 180 / np.pi
            a21 = np.arccos(_dot2(v1n, d2)) * 180 / np.pi
            a22 = np.arccos(_dot2(v2n, d2)) * 180 / np.pi
        # python's min()/max() on floats: min(a, b) = b if b < a else a
        t1 = np.where(a21 < a11, a21, a11)
        t2 = np.where(a22 < a12, a22, a12)
        tmin = np.where(t2 < t1, t2, t1)
        tmax = np.where(t2 > t1, t2, t1)
        return d, th_v, tmin, tmax
    raise ValueError(mode)


def edge_features(X: np.ndarray, V: np.ndarray, E: np.ndarray, features: Sequence[str], edge_mode: str) -> np.ndarray:
    """``GeometricGraph.extract_node_pair_features`` (graph.py:139-223) -> float64 [E, De]."""
    for f in features:
        if f not in EDGE_FEATURE_WIDTH:
            raise Exception("Invalid feature specified")                      # graph.py:219-220
    i = E[:, 0].astype(np.int64)
    j = E[:, 1].astype(np.int64)
    Xi, Xj, Vi, Vj = X[i, :2], X[j, :2], V[i, :2], V[j, :2]
    cols = []
    for f in features:
        if f == "point_pair_features":                                        # graph.py:183-186
            cols.extend(point_pair_features(Xi, Xj, Vi, Vj, edge_mode))
        elif f == "spatial_euclidean_distance":                               # graph.py:188-190
            t = Xi - Xj
            cols.append(np.sqrt(t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1]))
        elif f == "velocity_euclidean_distance":                              # graph.py:192-194
            t = Vi - Vj
            cols.append(np.sqrt(t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1]))
        elif f == "relative_position":                                        # graph.py:196-206
            t = Xi - Xj
            cols.extend([t[:, 0], t[:, 1]] if edge_mode == "directed" else [np.abs(t[:, 0]), np.abs(t[:, 1])])
        elif f == "relative_velocity":                                        # graph.py:208-216
            t = Vi - Vj
            cols.extend([t[:, 0], t[:, 1]] if edge_mode == "directed" else [np.abs(t[:, 0]), np.abs(t[:, 1])])
    if not cols:
        return np.empty((E.shape[0], 0))
    return np.stack(cols, axis=1).astype(np.float64)


def time_index(timestamp: np.ndarray) -> np.ndarray:
    """dataset_creation.py:214-223: rank of each timestamp among the frame's sorted unique values."""
    _, inv = np.unique(timestamp.reshape(-1), return_inverse=True)
    return inv.reshape(-1, 1).astype(np.float64)


def node_features(X, V, F: dict, degree: Optional[np.ndarray], features: Sequence[str]) -> np.ndarray:
    """``GeometricGraph.extract_single_node_features`` (graph.py:225-275) -> float64 [N, Dn]."""
    n = X.shape[0]
    cols = []
    for f in features:
        if f == "rcs":
            cols.append(np.asarray(F["rcs"], dtype=np.float64).reshape(n, 1))
        elif f == "time_index":
            cols.append(np.asarray(F["time_index"], dtype=np.float64).reshape(n, 1))
        elif f == "degree":
            cols.append(np.asarray(degree, dtype=np.float64).reshape(n, 1))
        elif f == "velocity_vector_length":
            cols.append(np.sqrt(V[:, 0] * V[:, 0] + V[:, 1] * V[:, 1]).reshape(n, 1))
        elif f == "velocity_vector":
            cols.append(np.asarray(V, dtype=np.float64))
        elif f == "spatial_coordinates":
            cols.append(np.asarray(X, dtype=np.float64))
        else:
            raise KeyError(f)
    return np.concatenate(cols, axis=1)


# --------------------------------------------------------------------------------------------
# the caller: GraphConstructor.build_geometric_graph + create_graph_data
# --------------------------------------------------------------------------------------------
def build_frame_graph(X_cc, V_cc, rcs, timestamp, algorithm: str, k: Optional[int], r: Optional[float],
                      node_feature_names: Sequence[str], edge_feature_names: Sequence[str], edge_mode: str,
                      distance_definition: str = "X") -> dict:
    """``GraphConstructor.build_geometric_graph`` (radarscenes/dataset_creation.py:190-229) followed by the
    dtype hand-off of ``create_graph_data`` (:786-814).  Returns the float64 intermediates and the
    tensors-to-be: ``x`` f32 [N,Dn], ``edge_index`` int64 [2,E], ``edge_attr`` f32 [E,De]."""
    basis = X_cc if distance_definition == "X" else np.concatenate((X_cc, V_cc), axis=1)
    F = {"rcs": rcs}
    if "time_index" in node_feature_names:
        F["time_index"] = time_index(timestamp)
    E = build_edges(basis, algorithm, k=k, r=r)
    n = X_cc.shape[0]
    if E is None:
        raise ValueError("graph with <= 1 node has no edge list (reference leaves E = None)")
    E_feat = edge_features(X_cc, V_cc, E, edge_feature_names, edge_mode)
    deg = undirected_degree(E, n) if "degree" in node_feature_names else None
    X_feat = node_features(X_cc, V_cc, F, deg, node_feature_names)
    return {
        "E": E, "E_feat": E_feat, "X_feat": X_feat, "degree": deg,
        "x": X_feat.astype(np.float32), "edge_index": np.ascontiguousarray(E.T.astype(np.int64)),
        "edge_attr": E_feat.astype(np.float32),
    }


def collate(graphs: List[dict]) -> dict:
    """PyG ``Batch`` collation used by ``utils/data_handling.py:30``: node/edge tensors concatenated on
    dim 0, ``edge_index`` on dim 1 with cumulative node offsets."""
    off = 0
    xs, eis, eas = [], [], []
    for g in graphs:
        xs.append(g["x"])
        eis.append(g["edge_index"] + off)
        eas.append(g["edge_attr"])
        off += g["x"].shape[0]
    return {"x": np.concatenate(xs), "edge_index": np.concatenate(eis, axis=1), "edge_attr": np.concatenate(eas)}
