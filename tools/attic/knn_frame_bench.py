"""k_knn_frame (brute force per small frame) against the grid walk (k_knn_team) on the C3 batch (512 x 300 points, k = 20) and on
ragged small frames: time per batch, equal rows (tools only).   python tools/attic/knn_frame_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from radargnn_amd import ops, synthetic


def run(frames, k, label):
    cat, ptr = synthetic.concat_frames(frames)
    X = torch.from_numpy(cat.X).cuda(); P = torch.from_numpy(ptr).cuda()
    big = max(f.n for f in frames)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for name, mf in (("grid walk", 0), ("per frame", big)):
        st = {}
        for _ in range(3):
            out = ops.knn_graph(X, P, k, static=st, max_frame_points=mf)
        torch.cuda.synchronize()
        # (the grid build is part of both: time the search launch alone through a second timing without it)
        e0.record()
        for _ in range(20):
            out = ops.knn_graph(X, P, k, static=st, max_frame_points=mf)
        e1.record(); torch.cuda.synchronize()
        t_all = e0.elapsed_time(e1) / 20
        g = st["grid"]
        e0.record()
        for _ in range(20):
            g.build(cell_size=0.0, pts_per_cell=3.0 if k >= 16 else 2.0, max_frame_points=mf)
        e1.record(); torch.cuda.synchronize()
        t_grid = e0.elapsed_time(e1) / 20
        res[name] = (out[0].clone(), (t_all - t_grid) * 1e3, t_grid * 1e3)
    same = torch.equal(res["grid walk"][0], res["per frame"][0])
    print(f"{label}: N={cat.X.shape[0]} k={k}: grid walk {res['grid walk'][1]:7.1f} us  per frame {res['per frame'][1]:7.1f} us  "
          f"(grid build {res['per frame'][2]:.1f} us)  rows equal: {same}")


if __name__ == "__main__":
    run([synthetic.nuscenes_frame(i) for i in range(512)], 20, "C3 512 x 300")
    run([synthetic.nuscenes_frame(i) for i in range(512)], 10, "512 x 300")
    run([synthetic.small_frame(500, i) for i in range(256)], 20, "256 x 500")
    run([synthetic.small_frame(1000, i) for i in range(128)], 20, "128 x 1000")
