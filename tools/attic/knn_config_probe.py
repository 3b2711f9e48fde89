import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from radargnn_amd import frames as fr, synthetic
which = sys.argv[1]
if which == "c3":
    frames = [synthetic.nuscenes_frame(i) for i in range(512)]; model = bench.shipped_model([224, 224, 128, 64, 32], 11)
else:
    frames = [synthetic.radarscenes_frame(i) for i in range(64)]; model = bench.shipped_model([224, 224, 128, 64, 32], 6)
model = model.cuda()
batch = fr.FrameBatch.from_frames(frames)
hot = fr.HotPath(model, fr.GraphSettings(algorithm="knn", k=20), use_hip_graphs=True)
for _ in range(5): hot(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): hot(batch)
torch.cuda.synchronize(); print(which, os.environ.get("RGNN_NO_CSR_FRAMES"), (time.perf_counter() - t0) / 20 * 1e3, "ms")
