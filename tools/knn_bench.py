"""A/B of the kNN search variants (RGNN_KNN_TEAM = 0 one thread per query, 16 / 32 / 64 lanes per query) on the C1, C3 and
C4 shapes: time of rgnn_knn_graph alone (grid already built) and bit-for-bit comparison with the first variant.
    python tools/knn_bench.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radargnn_amd import frames as fr, ops, synthetic
from radargnn_amd._lib import lib
import ctypes as C


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    cases = [("C1 1 x 3000, k=10", [synthetic.radarscenes_frame(0)], 10),
             ("C3 512 x 300, k=20", [synthetic.nuscenes_frame(i) for i in range(512)], 20),
             ("C4 64 x 3000, k=20", [synthetic.radarscenes_frame(i) for i in range(64)], 20),
             ("64 x 3000, k=1", [synthetic.radarscenes_frame(i) for i in range(64)], 1),
             ("64 x 3000, k=40", [synthetic.radarscenes_frame(i) for i in range(64)], 40)]
    variants = ["0", "16", "32", "64"]
    for name, frames, k in cases:
        b = fr.FrameBatch.from_frames(frames)
        X = b.X
        g = ops.GridHash(X, b.frame_ptr).build(cell_size=0.0, pts_per_cell=2.0)
        n = g.n
        outs, times = [], {v: [] for v in variants}
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        for v in variants:
            os.environ["RGNN_KNN_TEAM"] = v
            __import__("radargnn_amd").ops.reload_env()
            nbr = torch.full((n, k), -7, dtype=torch.int32, device="cuda")
            ei = torch.full((2, n * k), -7, dtype=torch.int64, device="cuda")
            ops.check(lib.rgnn_knn_graph(C.byref(g.desc), k, ops._ptr(nbr), ops._ptr(ei), ops._ptr(st), ops._stream()))
            torch.cuda.synchronize()
            outs.append((nbr, ei))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        nbr = torch.empty((n, k), dtype=torch.int32, device="cuda")
        for _ in range(rounds):
            for v in variants:
                os.environ["RGNN_KNN_TEAM"] = v
                __import__("radargnn_amd").ops.reload_env()
                lib.rgnn_knn_graph(C.byref(g.desc), k, ops._ptr(nbr), None, ops._ptr(st), ops._stream())
                e0.record()
                for _ in range(5):
                    lib.rgnn_knn_graph(C.byref(g.desc), k, ops._ptr(nbr), None, ops._ptr(st), ops._stream())
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 5)
        print(f"{name}: N={n}")
        for j, v in enumerate(variants):
            t = sorted(times[v])
            bad = int((outs[j][0] != outs[0][0]).sum()) + int((outs[j][1] != outs[0][1]).sum())
            print(f"    team {v:>2s}: {t[len(t) // 2] * 1e3:8.1f} us (min {t[0] * 1e3:8.1f})  mismatches vs one-thread kernel {bad}")
    os.environ.pop("RGNN_KNN_TEAM", None)
    __import__("radargnn_amd").ops.reload_env()


if __name__ == "__main__":
    main()
