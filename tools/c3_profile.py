"""One C3 batch (512 nuScenes-shaped frames x 300 points, kNN k = 20, shipped 5-layer model, 11 classes) for a kernel profile:
    rocprofv3 --kernel-trace --stats -- python tools/c3_profile.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = bench.shipped_model([224, 224, 128, 64, 32], 11).cuda()
batch = fr.FrameBatch.from_frames([synthetic.nuscenes_frame(i) for i in range(512)])
hot = fr.HotPath(model, fr.GraphSettings(algorithm="knn", k=20))
for _ in range(steps):
    hot(batch)
torch.cuda.synchronize()
