"""Where the window kernel starts to pay on r = 1 m batches: captured C2-model step, per-edge kernel against window kernel, by batch size
(tools only).

    python tools/win_threshold_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic
from radargnn_amd.gnn import mpnn_layers


def run(model, frames, use):
    mpnn_layers.USE_WINDOW_KERNEL = use
    batch = fr.FrameBatch.from_frames(frames)
    hot = fr.HotPath(model, bench.c2_settings(), use_hip_graphs=True)
    for _ in range(4):
        out = hot(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        hot(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 30 * 1e3, int(out[2].edge_index.shape[1])


model = bench.c2_model().cuda()
mpnn_layers_min = 1 << 18
for nf in (4, 8, 12, 16, 24, 32, 48, 64):
    frames = [synthetic.radarscenes_frame(i) for i in range(nf)]
    res = {}
    for use in (False, True, False, True):
        # the rule's edge threshold is part of wants_window_kernel: overridden here by patching the method's constant through the env-free path
        t, e = run(model, frames, use)
        res.setdefault(use, []).append(t)
    print(f"{nf:3d} frames, {e:7d} edges: per-edge {min(res[False]):.3f} ms, window {min(res[True]):.3f} ms  x{min(res[False]) / min(res[True]):.3f}", flush=True)
