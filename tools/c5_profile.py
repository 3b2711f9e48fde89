"""One C5 step (100k-point cloud, radius r = 1, 6-layer model on rotation-invariant features) for a kernel profile:
    rocprofv3 --kernel-trace --stats -- python tools/c5_profile.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = bench.shipped_model([224, 224, 224, 128, 64, 32], 6, node_dim=4, edge_dim=4).cuda()
batch = fr.FrameBatch.from_frames([synthetic.stress_cloud()])
hot = fr.HotPath(model, fr.GraphSettings(algorithm="radius", r=1.0,
                                         node_features=("rcs", "velocity_vector_length", "time_index", "degree"),
                                         edge_features=("point_pair_features",)))
for _ in range(steps):
    hot(batch)
torch.cuda.synchronize()
