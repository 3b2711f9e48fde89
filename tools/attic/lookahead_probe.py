"""Resident multi-batch loops: hot(batch) one after the other against HotPath.begin / finish with the next batch's search a batch ahead
on the side stream (kNN: the whole search runs beside the previous batch's model kernels).  C3 / C4 shapes, 4 resident batches.
    python tools/attic/lookahead_probe.py [c4|c3] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from radargnn_amd import frames as fr, synthetic

case = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = fr.GraphSettings(algorithm="knn", k=20)
if case == "c4":
    model = bench.shipped_model([224, 224, 128, 64, 32], 6).cuda()
    bs = [fr.FrameBatch.from_frames([synthetic.radarscenes_frame(64 * b + i) for i in range(64)]) for b in range(4)]
else:
    model = bench.shipped_model([224, 224, 128, 64, 32], 11).cuda()
    bs = [fr.FrameBatch.from_frames([synthetic.nuscenes_frame(512 * b + i) for i in range(512)]) for b in range(4)]
hot = fr.HotPath(model, cfg)


def loop(fn, n):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


state = {"i": 0}


def plain():
    hot(bs[state["i"] % 4]); state["i"] += 1


plain_ms = loop(plain, steps)
st = {"ahead": hot.begin(bs[0]), "i": 1}


def halves():
    nxt = hot.begin(bs[st["i"] % 4])
    hot.finish(st["ahead"])
    st["ahead"], st["i"] = nxt, st["i"] + 1


halves_ms = loop(halves, steps)
hot.finish(st["ahead"])
torch.cuda.synchronize()
print(case, {"one after the other": round(plain_ms, 4), "search a batch ahead": round(halves_ms, 4)})
