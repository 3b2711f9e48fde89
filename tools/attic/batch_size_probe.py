"""Per-frame step time of the C2 workload at different batch sizes (is a batch whose intermediates fit the 256 MB
infinity cache faster per frame?):  python tools/attic/batch_size_probe.py [frames ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic

sizes = [int(a) for a in sys.argv[1:]] or [64, 32, 16, 8]
model = bench.c2_model().cuda()
cfg = fr.GraphSettings(algorithm="radius", r=1.0)
for b in sizes:
    batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(b)])
    for mode in (False, True):
        hot = fr.HotPath(model, cfg, use_hip_graphs=mode)
        with torch.no_grad():
            for _ in range(6):
                hot(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                hot(batch)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"frames {b:3d} graph={mode}: {dt * 1e3:7.3f} ms per batch, {dt * 1e6 / b:7.2f} us per frame, {b / dt:9.0f} frames/s", flush=True)
