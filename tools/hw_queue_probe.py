"""Which HIP streams share a hardware queue?  (ROCm maps the streams of a process onto GPU_MAX_HW_QUEUES = 4 queues per priority level;
two streams on one queue run one after the other whatever their events say.)
A spinning kernel occupies stream A; a one-element fill behind it on stream B completes at once iff B sits on another queue.
    python tools/hw_queue_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def independent(a, b, x):
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(40_000_000)                         # ~20 ms
    ev = torch.cuda.Event()
    with torch.cuda.stream(b):
        x.fill_(1.0)
        ev.record(b)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.004 and not ev.query():
        pass
    ok = ev.query()
    torch.cuda.synchronize()
    return ok


def main():
    try:
        print("priority range:", torch.cuda.Stream.priority_range())
    except Exception as e:                                    # noqa: BLE001
        print("priority range: n/a", e)
    x = torch.zeros(16, device="cuda")
    streams = {"default": torch.cuda.default_stream()}
    for i in range(6):
        streams[f"n{i}"] = torch.cuda.Stream()
    for i in range(6):
        streams[f"h{i}"] = torch.cuda.Stream(priority=-1)
    for pr in (1, 2):
        try:
            streams[f"p{pr}"] = torch.cuda.Stream(priority=pr)
        except Exception as e:                                # noqa: BLE001
            print(f"priority {pr}: refused ({e})")
    names = list(streams)
    print("stream ids:", {k: (v.stream_id, v.priority) for k, v in streams.items()})
    print("rows: busy stream; columns: stream that tries to run beside it; X = waits (same hardware queue)")
    print("        " + " ".join(f"{n:>7}" for n in names))
    for a in names:
        row = []
        for b in names:
            row.append("   -   " if a == b else ("   .   " if independent(streams[a], streams[b], x) else "   X   "))
        print(f"{a:>7} " + " ".join(row), flush=True)


if __name__ == "__main__":
    main()
