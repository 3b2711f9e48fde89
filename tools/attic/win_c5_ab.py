"""A/B of the window kernel on the C5 stress cloud (100 000 points, radius 1 m, 6-layer model): ms per step with / without it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from radargnn_amd import frames as fr, synthetic
from radargnn_amd.gnn import mpnn_layers
model = bench.shipped_model([224, 224, 224, 128, 64, 32], 6, node_dim=4, edge_dim=4).cuda()
cfg = fr.GraphSettings(algorithm="radius", r=1.0, node_features=("rcs", "velocity_vector_length", "time_index", "degree"), edge_features=("point_pair_features",))
batch = fr.FrameBatch.from_frames([synthetic.stress_cloud()])
for use in (False, True, False, True):
    mpnn_layers.USE_WINDOW_KERNEL = use
    hot = fr.HotPath(model, cfg, use_hip_graphs=True)
    for _ in range(4): c, b, g = hot(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hot(batch)
    torch.cuda.synchronize()
    print("window" if use else "per-edge", round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms; edges", g.edge_index.shape[1], flush=True)
