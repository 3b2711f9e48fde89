"""Step times of the BASELINE configurations with whatever build of librgnn RGNN_LIB names (default: the in-tree one): for A/B runs of
kernel variants on ONE box in ONE gpurun call (the pool's boxes differ by several per cent).
    python tools/ab_step.py [--cases c2,c3,c4,c5] [--steps 30]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic


def timed(hot, batches, steps):
    for _ in range(4):
        for b in batches:
            hot(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for b in batches:
            hot(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps / len(batches) * 1e3


def main():
    argv = sys.argv[1:]
    cases, steps, kernels = ["c2", "c3", "c4"], 30, False
    while argv:
        a = argv.pop(0)
        if a == "--cases": cases = argv.pop(0).split(",")
        elif a == "--steps": steps = int(argv.pop(0))
        elif a == "--kernels": kernels = True
    out = {}
    rs = lambda a, b: [synthetic.radarscenes_frame(i) for i in range(a, b)]
    for c in cases:
        if c == "c2":
            model, cfg, fb, graph = bench.c2_model().cuda(), bench.c2_settings(), [rs(0, 64)], True
        elif c == "c3":
            model, cfg, fb, graph = bench.shipped_model([224, 224, 128, 64, 32], 11).cuda(), fr.GraphSettings(algorithm="knn", k=20), \
                [[synthetic.nuscenes_frame(i) for i in range(512)]], True
        elif c == "c4":
            model, cfg, fb, graph = bench.shipped_model([224, 224, 128, 64, 32], 6).cuda(), fr.GraphSettings(algorithm="knn", k=20), [rs(0, 64)], True
        else:
            model = bench.shipped_model([224, 224, 224, 128, 64, 32], 6, node_dim=4, edge_dim=4).cuda()
            cfg = fr.GraphSettings(algorithm="radius", r=1.0, node_features=("rcs", "velocity_vector_length", "time_index", "degree"),
                                   edge_features=("point_pair_features",))
            fb, graph = [[synthetic.stress_cloud()]], True
        batches = [fr.FrameBatch.from_frames(f) for f in fb]
        out[c] = round(timed(fr.HotPath(model, cfg, use_hip_graphs=graph), batches, steps), 4)
        if kernels:                                     # HIP events inside librgnn around the edge / dense launches (eager pass)
            summ = bench.instrumented(model, cfg, batches[:1], 5, c in ("c2", "c5"))
            roof, gather = bench.rooflines(summ, 5, with_pmc=False)
            out[c + "_edge_us"] = round(gather["avg_launch_ms"] * 1e3, 1) if gather else None
            out[c + "_dense_us"] = round(roof["avg_launch_ms"] * 1e3, 1) if roof else None
    print(os.environ.get("RGNN_LIB", "in-tree"), out, flush=True)


if __name__ == "__main__":
    main()
