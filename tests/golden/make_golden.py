"""Generate the golden vectors of tests/golden/*.npz by IMPORTING the reference's graph constructor.

Runs only in the build container (it needs /root/reference); the GPU box and the test-suite only
read the committed .npz files.  Nothing of the reference's code is stored: the fixtures hold the
synthetic inputs (radargnn_amd.synthetic) and the arrays the reference produced for them.

    python tests/golden/make_golden.py

Reference entry points exercised (src/gnnradarobjectdetection/graph_constructor/graph.py):
``GeometricGraph.build`` (:32), ``extract_node_pair_features`` (:139), ``extract_single_node_features``
(:225), ``get_degree`` (:93).  networkx >= 3 dropped ``from_numpy_matrix`` (used at graph.py:94), so it is
aliased to ``from_numpy_array`` in this process only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference/src")          # the reference package wins the name
sys.path.append(REPO)                              # radargnn_amd.synthetic

import networkx as nx  # noqa: E402

if not hasattr(nx, "from_numpy_matrix"):
    nx.from_numpy_matrix = nx.from_numpy_array

import gnnradarobjectdetection.graph_constructor.graph as ref_graph  # noqa: E402

assert ref_graph.__file__.startswith("/root/reference/"), ref_graph.__file__
from radargnn_amd import synthetic  # noqa: E402

ALL_EDGE = ["point_pair_features", "spatial_euclidean_distance", "velocity_euclidean_distance",
            "relative_position", "relative_velocity"]
ALL_NODE = ["rcs", "time_index", "degree", "velocity_vector_length", "velocity_vector", "spatial_coordinates"]


def time_index_ref(timestamp):
    """The loop of preprocessor/radarscenes/dataset_creation.py:214-223 run on its own inputs (that
    module itself imports ray/radar_scenes, which are absent here)."""
    stamps = np.unique(timestamp)
    t_idx = np.zeros_like(timestamp)
    for i, _ in enumerate(stamps):
        t_idx[np.where(timestamp == stamps[i])[0]] = int(i)
    return t_idx


def run_reference(frame, routine, k, r, edge_features, edge_mode, node_features, basis="X"):
    g = ref_graph.GeometricGraph()
    g.X, g.V = frame.X, frame.V
    g.F = {"rcs": frame.rcs}
    g.add_invariant_feature("time_index", time_index_ref(frame.timestamp))
    dist_basis = frame.X if basis == "X" else np.concatenate((frame.X, frame.V), axis=1)
    g.build(dist_basis, routine, k=k, r=r)
    g.extract_node_pair_features(edge_features, edge_mode)
    g.extract_single_node_features(node_features)
    return g


def pack(frame, g, extra=None):
    d = dict(X=frame.X, V=frame.V, rcs=frame.rcs, timestamp=frame.timestamp,
             E=g.E.astype(np.int32), degree=np.asarray(g.F["degree"]).reshape(-1).astype(np.int32),
             time_index=np.asarray(g.F["time_index"]).reshape(-1))
    if extra:
        d.update(extra)
    return d


def main():
    out = {}
    # ---- small frames: every feature kind x {directed, undirected}, kNN and radius, with
    # zero-velocity points and exact duplicate points
    cases = []
    for n, seed, dup in [(6, 0, 0), (6, 1, 1), (40, 2, 2), (300, 3, 0)]:
        fr = synthetic.small_frame(n, seed, duplicates=dup) if n < 300 else synthetic.nuscenes_frame(0)
        for routine, k, r in [("knn", 1, 1), ("knn", min(10, n // 2 - 1), 1), ("radius", 1, 2.5 if n < 300 else 6.0)]:
            for mode in ("directed", "undirected"):
                for basis in (("X", "XV") if (n == 40 and routine == "knn") else ("X",)):
                    cases.append((fr, n, seed, routine, k, r, mode, basis))
    for fr, n, seed, routine, k, r, mode, basis in cases:
        with np.errstate(all="ignore"):
            g = run_reference(fr, routine, k, r, ALL_EDGE, mode, ALL_NODE, basis)
        name = f"small_n{n}_s{seed}_{routine}_k{k}_r{r}_{mode}_{basis}"
        d = pack(fr, g, dict(E_feat=g.E_feat, X_feat=g.X_feat))
        d["meta"] = np.array([routine, str(k), str(r), mode, basis])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, g.E.shape, g.E_feat.shape, g.X_feat.shape)

    # ---- RadarScenes-shaped 3000-point frame: topology for the shipped and the bench settings,
    # the shipped feature set (translation invariant), float32 like create_graph_data hands them over
    fr = synthetic.radarscenes_frame(0)
    node_feats = ["rcs", "velocity_vector", "time_index", "degree"]
    for routine, k, r in [("knn", 10, 1), ("knn", 20, 1), ("radius", 1, 1.0)]:
        g = run_reference(fr, routine, k, r, ["relative_position"], "directed", node_feats)
        name = f"rs3000_{routine}_k{k}_r{r}"
        d = pack(fr, g, dict(E_feat=g.E_feat.astype(np.float32), X_feat=g.X_feat.astype(np.float32)))
        d["meta"] = np.array([routine, str(k), str(r), "directed", "X"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, g.E.shape, "deg", d["degree"].min(), d["degree"].max())

    # ---- rotation-invariant feature set on a 3000-point radius graph (point-pair features at scale)
    g = run_reference(fr, "radius", 1, 1.0, ["point_pair_features"], "directed",
                      ["rcs", "velocity_vector_length", "time_index", "degree"])
    d = pack(fr, g, dict(E_feat=g.E_feat.astype(np.float32), X_feat=g.X_feat.astype(np.float32)))
    d["meta"] = np.array(["radius", "1", "1.0", "directed", "X"])
    np.savez_compressed(os.path.join(HERE, "rs3000_radius_ppf.npz"), **d)
    print("rs3000_radius_ppf", g.E.shape)


if __name__ == "__main__":
    main()
