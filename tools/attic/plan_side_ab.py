"""Window plan on the side stream (TargetCSR.start_win_plan) against the plan in line, C4 batch and C3, captured steps (tools only).

    python tools/attic/plan_side_ab.py
"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic
from radargnn_amd.gnn import mpnn_layers


def run(frames, model, side):
    mpnn_layers.PLAN_ON_SIDE_STREAM = side
    batch = fr.FrameBatch.from_frames(frames)
    hot = fr.HotPath(model, fr.GraphSettings(algorithm="knn", k=20), use_hip_graphs=True)
    for _ in range(4):
        hot(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        hot(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


for name, frames, k in (("C4 batch", [synthetic.radarscenes_frame(i) for i in range(64)], 6),
                        ("C3", [synthetic.nuscenes_frame(i) for i in range(512)], 11)):
    model = bench.shipped_model([224, 224, 128, 64, 32], k).cuda()
    for side in (False, True, False, True, False, True):
        print(f"{name}: plan {'on the side stream' if side else 'in line          '} {run(frames, model, side):.3f} ms per captured step", flush=True)
