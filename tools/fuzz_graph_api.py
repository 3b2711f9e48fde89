"""Randomised check of the single-graph surface (radargnn_amd.graph_constructor.build_geometric_graph -- the reference's
GraphConstructor.build_geometric_graph, radarscenes/dataset_creation.py:190-229) against the CPU oracle: random point clouds (2 ...
2000 points, exact duplicates, zero velocities, collinear points, equal timestamps), radius / kNN, X / XV distances, every node and
edge feature list, directed / undirected.  Edge lists bit-equal, float64 features within 1e-12 relative (device libm vs numpy).
    python tools/fuzz_graph_api.py [cases] [seed]        (test infrastructure: imports oracle/)"""
import os, sys
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import graph_oracle as go
from radargnn_amd.graph_constructor import GraphConstructionConfiguration, build_geometric_graph

NODE = ["rcs", "velocity_vector", "time_index", "degree", "velocity_vector_length", "spatial_coordinates"]
EDGE = ["relative_position", "point_pair_features", "spatial_euclidean_distance", "velocity_euclidean_distance", "relative_velocity"]


def one(rng, case):
    n = int(rng.choice([2, 3, 7, 60, 400, 2000]))
    X = rng.uniform(-50, 50, (n, 2)) if rng.random() < 0.7 else np.round(rng.uniform(-5, 5, (n, 2)), 0)      # (grid points: ties in distance)
    V = rng.normal(0, 3, (n, 2))
    if rng.random() < 0.4:
        V[rng.random(n) < 0.4] = 0.0
    if rng.random() < 0.3 and n > 3:
        X[1::3] = X[0]; V[1::3] = V[0] if rng.random() < 0.5 else V[1::3]
    if rng.random() < 0.15:
        X[:, 1] = 2.0 * X[:, 0] + 1.0                          # collinear
    rcs = rng.normal(0, 5, (n, 1)); ts = rng.choice(rng.uniform(0, 1, int(rng.integers(1, 8))), (n, 1))
    algo = "knn" if rng.random() < 0.5 else "radius"
    k = int(rng.choice([1, 2, 5, 20])); r = float(rng.choice([0.3, 1.0, 4.0, 200.0]))
    if algo == "knn":
        k = min(k, n - 1)
    if algo == "radius" and r > 50 and n > 400:
        r = 4.0                                                 # (keeps the oracle's dense adjacency small)
    nodes = [NODE[i] for i in rng.permutation(6)[:int(rng.integers(1, 7))]]
    edges = [EDGE[i] for i in rng.permutation(5)[:int(rng.integers(1, 6))]]
    mode = "directed" if rng.random() < 0.5 else "undirected"; dist = "X" if rng.random() < 0.6 else "XV"
    desc = f"case {case}: n {n} {algo} k {k} r {r} {mode} {dist} nodes {nodes} edges {edges}"
    if os.environ.get("FUZZ_VERBOSE"):
        print(desc, flush=True)
    pc = SimpleNamespace(X_cc=X, V_cc_compensated=V, rcs=rcs, timestamp=ts)
    cfg = GraphConstructionConfiguration(algo, {"k": k, "r": r}, nodes, edges, mode, dist)
    g = build_geometric_graph(cfg, pc)
    ref = go.build_frame_graph(X, V, rcs, ts, algo, k, r, nodes, edges, mode, dist)
    bad = []
    E = np.asarray(g.E)
    if E.shape != ref["E"].shape or not np.array_equal(E, ref["E"]):
        bad.append(f"E {E.shape} vs {ref['E'].shape}")
    else:
        if not np.allclose(np.asarray(g.X_feat, dtype=np.float64), ref["X_feat"], rtol=1e-12, atol=1e-12):
            bad.append("X_feat")
        ef, rf = np.asarray(g.E_feat, dtype=np.float64), ref["E_feat"]
        if ef.shape != rf.shape or not np.allclose(ef, rf, rtol=1e-9, atol=1e-9, equal_nan=True):
            bad.append(f"E_feat max diff {np.nanmax(np.abs(ef - rf)) if ef.shape == rf.shape and ef.size else 'shape'}")
    return ("FAIL " + desc + " -> " + ", ".join(bad)) if bad else "ok"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fails = 0
    for c in range(cases):
        try:
            r = one(rng, c)
        except Exception as e:                                  # noqa: BLE001
            r = f"FAIL case {c}: {type(e).__name__}: {str(e)[:300]}"
        if r != "ok":
            print(r, flush=True); fails += 1
    print(f"{cases} cases, {fails} failures")


if __name__ == "__main__":
    main()
