"""Host time of ONE eager C2 step by function (cProfile, main thread): what the streaming path (frames.FrameStreamer) pays per batch.

    python tools/host_profile_eager.py [n]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic

model = bench.c2_model().cuda()
hot = fr.HotPath(model, bench.c2_settings(), use_hip_graphs=False)
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(int(os.environ.get("NF", "64")))])
for _ in range(5):
    hot(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    hot(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager step: host returns after {(t1 - t0) / 20 * 1e3:.3f} ms, device done after {(t2 - t0) / 20 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    hot(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 28)
