import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from radargnn_amd import frames as fr, synthetic
from radargnn_amd.gnn import mpnn_layers
def run(name, frames, model, use):
    mpnn_layers.USE_WINDOW_KERNEL = use
    batch = fr.FrameBatch.from_frames(frames)
    out = {}
    for mode in (False, True):
        hot = fr.HotPath(model, fr.GraphSettings(algorithm="knn", k=20), use_hip_graphs=mode)
        for _ in range(4): hot(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): c, b, g = hot(batch)
        torch.cuda.synchronize(); out[mode] = (time.perf_counter() - t0) / 15 * 1e3
    return out, c
for name, frames, k in (("C4 batch", [synthetic.radarscenes_frame(i) for i in range(64)], 6), ("C3", [synthetic.nuscenes_frame(i) for i in range(512)], 11)):
    model = bench.shipped_model([224, 224, 128, 64, 32], k).cuda()
    res = {}
    for use in (False, True, False, True):
        t, c = run(name, frames, model, use)
        res.setdefault(use, []).append(t); 
        print(name, "window kernel" if use else "per-edge kernel", {("graph" if m else "eager"): round(v, 3) for m, v in t.items()}, flush=True)
