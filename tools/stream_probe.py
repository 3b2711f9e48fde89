"""Where the streamed C2 step (frames.FrameStreamer) loses against the resident one: ms per batch of
  captured  -- one HIP graph per step on a resident batch (bench.py's `value`)
  eager     -- eager launches on a resident batch (one host read of the edge count in the middle of a step)
  halves    -- HotPath.begin / finish one batch ahead on four resident batches (no copies, no loader thread)
  stream    -- the full FrameStreamer (loader thread, pinned staging, H2D / D2H)
  stream-nostage -- the same with the loader's copies into the pinned block skipped (the block keeps the first batches' bytes)
    python tools/stream_probe.py [--steps 40]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from radargnn_amd import frames as fr, synthetic


def main():
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 40
    frames = [synthetic.radarscenes_frame(i) for i in range(64)]
    model, cfg = bench.c2_model().cuda(), bench.c2_settings()
    out = {}

    def loop(fn, n, name=None):
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        if name:
            out[name + "-host-returns"] = (t1 - t0) / n * 1e3
        return (time.perf_counter() - t0) / n * 1e3

    b0 = fr.FrameBatch.from_frames(frames)
    hot_g = fr.HotPath(model, cfg, use_hip_graphs=True)
    out["captured"] = loop(lambda: hot_g(b0), steps)
    hot = fr.HotPath(model, cfg, use_hip_graphs=False)
    out["eager"] = loop(lambda: hot(b0), steps)

    bs = [fr.FrameBatch.from_frames(frames) for _ in range(4)]
    state = {"ahead": hot.begin(bs[0]), "i": 1}

    def halves():
        nxt = hot.begin(bs[state["i"] % 4])
        hot.finish(state["ahead"])
        state["ahead"], state["i"] = nxt, state["i"] + 1
    out["halves"] = loop(halves, steps, "halves")
    hot.finish(state["ahead"])
    torch.cuda.synchronize()

    stalls = []

    def streamed(streamer):
        warm = 2 * streamer.slots
        stamps = []
        for _c, _b in streamer.run(frames for _ in range(warm + steps)):
            stamps.append(time.perf_counter())
        t = stamps[warm - 1:]
        raw = [(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])]
        gaps = sorted(raw)
        stalls.append([round(g, 2) for g in gaps[-3:]] + ["at", raw.index(gaps[-1]), "median", round(gaps[len(gaps) // 2], 3)])
        return (t[-1] - t[0]) / (len(t) - 1) * 1e3
    streamed(fr.FrameStreamer(hot))                           # (first-use allocations of the process: not a measurement)
    out["stream"] = streamed(fr.FrameStreamer(hot))
    out["stream-4slots-1behind"] = streamed(fr.FrameStreamer(hot, slots=4, behind=1))
    out["stream-again"] = streamed(fr.FrameStreamer(hot))
    out["stream-6slots-1behind"] = streamed(fr.FrameStreamer(hot, slots=6, behind=1))
    out["stream-8slots-3behind"] = streamed(fr.FrameStreamer(hot, slots=8, behind=3))
    out["stream-again2"] = streamed(fr.FrameStreamer(hot))
    out["stream-no-lookahead"] = streamed(fr.FrameStreamer(hot, lookahead=False))
    out["stream-again3"] = streamed(fr.FrameStreamer(hot))

    s2 = fr.FrameStreamer(hot)
    real = fr.ops.lib.rgnn_stage_frames
    calls = {"n": 0}

    def lazy_stage(*a):
        calls["n"] += 1
        return real(*a) if calls["n"] <= 2 * s2.slots else 0
    fr.ops.lib.rgnn_stage_frames = lazy_stage
    try:
        out["stream-nostage"] = streamed(s2)
    finally:
        fr.ops.lib.rgnn_stage_frames = real
    print({k: round(v, 4) for k, v in out.items()}, flush=True)
    print("longest three intervals + median per streamed run (ms):", stalls, flush=True)


if __name__ == "__main__":
    main()
