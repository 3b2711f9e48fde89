"""The model's public forward(x, edge_index, edge_attr) on the C2 batch (no symmetry / visiting-order hints: what a reference
script calls), for a kernel profile:  rocprofv3 --kernel-trace --stats -- python tools/attic/public_forward_profile.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
g = fr.build_graphs(batch, fr.GraphSettings(algorithm="radius", r=1.0))
with torch.no_grad():
    for _ in range(3):
        model(g.x, g.edge_index, g.edge_attr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(g.x, g.edge_index, g.edge_attr)
    torch.cuda.synchronize()
print(f"public forward: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per call", flush=True)
