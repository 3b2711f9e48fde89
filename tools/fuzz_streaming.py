"""Randomised check of frames.FrameStreamer against the resident path: random numbers of frames per batch (ragged batches, frames of a few
points, frames whose arrays need a conversion), radius / kNN graphs, random slots / behind / lookahead; every streamed batch must equal
HotPath on the same frames bit for bit, in order.
    python tools/fuzz_streaming.py [runs] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from radargnn_amd import frames as fr, gnn, synthetic


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    fails = 0
    for run in range(runs):
        algo = "radius" if rng.random() < 0.5 else "knn"
        k = int(rng.choice([3, 6]))
        cfg = fr.GraphSettings(algorithm=algo, k=k, r=float(rng.choice([1.5, 2.5])))
        torch.manual_seed(run)
        mcfg = gnn.GNNArchitectureConfig(5, 2, [int(rng.choice([32, 64, 224]))] * int(rng.integers(1, 3)), [6], [16, 5], True, bool(rng.random() < 0.5),
                                         [32, 64], [4, 8, 16], "MPNNConv", False)
        model = gnn.DetNetBasic(mcfg).cuda().eval()
        n_batches = int(rng.integers(3, 25))
        host = []
        for b in range(n_batches):
            nf = int(rng.integers(1, 12))
            fs = []
            for i in range(nf):
                f = synthetic.nuscenes_frame(int(rng.integers(0, 500))) if rng.random() < 0.7 else synthetic.radarscenes_frame(int(rng.integers(0, 50)), n_clusters=5, pts_per_cluster=12, n_clutter=60)
                n = int(rng.integers(k + 1, f.n + 1)) if rng.random() < 0.3 else f.n
                f = synthetic.RadarFrame(f.X[:n], f.V[:n], f.rcs[:n], f.timestamp[:n])
                if rng.random() < 0.1:                       # arrays that need a conversion on the way into the pinned block
                    f = synthetic.RadarFrame(np.asfortranarray(f.X), f.V.astype(np.float32), f.rcs, f.timestamp)
                fs.append(f)
            host.append(fs)
        hot = fr.HotPath(model, cfg)
        want = []
        for fs in host:
            c, b, g = hot(fr.FrameBatch.from_frames(fs))
            g.check()
            want.append((c.cpu(), b.cpu()))
        slots = int(rng.integers(2, 9))
        kw = dict(slots=slots, behind=int(rng.integers(1, 5)), lookahead=bool(rng.random() < 0.7))
        got = [(c.clone(), b.clone()) for c, b in fr.FrameStreamer(hot, **kw).run(iter(host))]
        ok = len(got) == len(want) and all(torch.equal(a[0], w[0]) and torch.equal(a[1], w[1]) for a, w in zip(got, want))
        if not ok:
            fails += 1
            print(f"FAIL run {run}: {algo} k {k} batches {n_batches} {kw}", flush=True)
    print(f"{runs} runs, {fails} failures")


if __name__ == "__main__":
    main()
