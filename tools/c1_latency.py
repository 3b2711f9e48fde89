"""C1 (one 3000-point frame, kNN k=10, 2-layer MPNNConv [224,224]): wall time per frame with eager launches and with
the HIP-graph replay, plus device-busy time per step.    python tools/c1_latency.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    model = bench.shipped_model([224, 224], 6).cuda()
    batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(0)])
    cfg = fr.GraphSettings(algorithm="knn", k=10)
    for mode in (False, True):
        hot = fr.HotPath(model, cfg, use_hip_graphs=mode)
        for _ in range(5):
            hot(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            hot(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(("graph" if mode else "eager") + f": {dt * 1e3:.3f} ms per frame", flush=True)


if __name__ == "__main__":
    main()
