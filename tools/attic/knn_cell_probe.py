"""Sweep of the grid's points-per-cell target for the kNN search (rgnn_knn_graph, team kernel) on the C1 / C3 / C4 shapes: time of the
grid build + search, neighbour lists compared bit for bit with the pts_per_cell = 2 result (tools only).
    python tools/attic/knn_cell_probe.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radargnn_amd import frames as fr, ops, synthetic
from radargnn_amd._lib import lib
import ctypes as C


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cases = [("C1 1 x 3000, k=10", [synthetic.radarscenes_frame(0)], 10),
             ("C3 512 x 300, k=20", [synthetic.nuscenes_frame(i) for i in range(512)], 20),
             ("C4 64 x 3000, k=20", [synthetic.radarscenes_frame(i) for i in range(64)], 20),
             ("64 x 3000, k=1", [synthetic.radarscenes_frame(i) for i in range(64)], 1),
             ("64 x 3000, k=5", [synthetic.radarscenes_frame(i) for i in range(64)], 5),
             ("64 x 3000, k=40", [synthetic.radarscenes_frame(i) for i in range(64)], 40)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, frames, k in cases:
        b = fr.FrameBatch.from_frames(frames)
        biggest = int(b.frame_sizes.max())
        ref = None
        print(f"{name}: N={b.num_points}")
        for ppc in (2.0, 3.0, 4.0, 6.0, 8.0, 12.0, 16.0, 24.0):
            st = torch.zeros(1, dtype=torch.int32, device="cuda")
            g = ops.GridHash(b.X, b.frame_ptr).build(cell_size=0.0, pts_per_cell=ppc, max_frame_points=biggest)
            n = g.n
            nbr = torch.full((n, k), -7, dtype=torch.int32, device="cuda")
            ops.check(lib.rgnn_knn_graph(C.byref(g.desc), k, ops._ptr(nbr), None, ops._ptr(st), ops._stream()))
            torch.cuda.synchronize()
            if ref is None:
                ref = nbr.clone()
            bad = int((nbr != ref).sum())
            ts, tg = [], []
            for _ in range(rounds):
                e0.record()
                for _ in range(5):
                    g.build(cell_size=0.0, pts_per_cell=ppc, max_frame_points=biggest)
                e1.record(); torch.cuda.synchronize(); tg.append(e0.elapsed_time(e1) / 5)
                e0.record()
                for _ in range(5):
                    lib.rgnn_knn_graph(C.byref(g.desc), k, ops._ptr(nbr), None, ops._ptr(st), ops._stream())
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
            ts.sort(); tg.sort()
            print(f"    pts_per_cell {ppc:5.1f}: search {ts[len(ts) // 2] * 1e3:8.1f} us, grid build {tg[len(tg) // 2] * 1e3:6.1f} us, "
                  f"mismatches vs pts_per_cell 2: {bad}, status {int(st.item())}")


if __name__ == "__main__":
    main()
