"""Throughput of the hot path on every BASELINE.json configuration (bench.py only times configs[1] = C2; the others
are parity-test cases).  One JSON line per configuration: frames/s with the batch resident in HBM, eager launches,
train-mode BatchNorm unless stated.  Results go to profiles/r01_config_bench.json via the caller."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, gnn, synthetic


def shipped(dims, k_classes, node_dim=5, edge_dim=2):
    return gnn.GNNArchitectureConfig(node_dim, edge_dim, dims, [k_classes], [16, 5], True, True, [32, 64, 128, 224],
                                     [4, 8, 16], "MPNNConv", False)


def run(name, model, settings, frames, steps, eval_mode=False, graphs=False):
    model = model.cuda()
    model.eval() if eval_mode else model.train()
    hot = fr.HotPath(model, settings, use_hip_graphs=graphs)
    batch = fr.FrameBatch.from_frames(frames)
    for _ in range(3):
        cls, bb, g = hot(batch)
    g.check()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hot(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    line = {"config": name, "frames": len(frames), "points": int(batch.num_points), "edges": int(g.edge_index.shape[1]),
            "ms_per_batch": dt * 1e3, "frames_per_s": len(frames) / dt, "points_per_s": batch.num_points / dt,
            "batchnorm": "eval" if eval_mode else "train", "launch_mode": "graph" if graphs else "eager"}
    print(json.dumps(line), flush=True)
    return line


def main():
    torch.manual_seed(0)
    out = []
    c1 = gnn.DetNetBasic(gnn.GNNArchitectureConfig(5, 2, [224, 224], [6], [16, 5], True, True, [32, 64, 128, 224], [4, 8, 16],
                                                   "MPNNConv", False))
    out.append(run("C1: 1 frame x 3000 pts, kNN k=10, 2-layer MPNNConv (latency case)", c1,
                   fr.GraphSettings(algorithm="knn", k=10), [synthetic.radarscenes_frame(0)], 50))
    out.append(run("C1 with the post-search stage replayed from a HIP graph", c1,
                   fr.GraphSettings(algorithm="knn", k=10), [synthetic.radarscenes_frame(0)], 50, graphs=True))
    c2 = bench.c2_model()
    rs64 = [synthetic.radarscenes_frame(i) for i in range(64)]
    out.append(run("C2: 64 frames x 3000 pts, radius r=1, 4-layer MPNNConv (bench.py workload)", c2,
                   fr.GraphSettings(algorithm="radius", k=0, r=1.0), rs64, 20))
    out.append(run("C2, eval-mode BatchNorm (running statistics)", c2, fr.GraphSettings(algorithm="radius", k=0, r=1.0),
                   rs64, 20, eval_mode=True))
    c3 = gnn.DetNetBasic(shipped([224, 224, 128, 64, 32], 11))
    out.append(run("C3: 512 frames x 300 pts, kNN k=20, shipped 5-layer model, 11 classes", c3,
                   fr.GraphSettings(algorithm="knn", k=20), [synthetic.nuscenes_frame(i) for i in range(512)], 20))
    c4 = gnn.DetNetBasic(shipped([224, 224, 128, 64, 32], 6))
    out.append(run("C4 (one rank's batch): 64 frames x 3000 pts, kNN k=20, shipped 5-layer model + both heads", c4,
                   fr.GraphSettings(algorithm="knn", k=20), rs64, 10))
    c5 = gnn.DetNetBasic(shipped([224, 224, 224, 128, 64, 32], 6, node_dim=4, edge_dim=4))
    out.append(run("C5: one 100k-point cloud, radius r=1 (~5 M edges), 6-layer model on rotation-invariant features", c5,
                   fr.GraphSettings(algorithm="radius", r=1.0,
                                    node_features=("rcs", "velocity_vector_length", "time_index", "degree"),
                                    edge_features=("point_pair_features",)), [synthetic.stress_cloud()], 10))
    return out


if __name__ == "__main__":
    main()
