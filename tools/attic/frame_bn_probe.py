import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from radargnn_amd import frames as fr, synthetic
model = bench.c2_model().cuda()
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
hot = fr.HotPath(model, bench.c2_settings(), use_hip_graphs=True, bn_scope="frame")
for _ in range(5): hot(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): hot(batch)
torch.cuda.synchronize(); print("per-frame BN C2", os.environ.get("RGNN_BN_SEG_SPLIT"), (time.perf_counter() - t0) / 30 * 1e3, "ms")
