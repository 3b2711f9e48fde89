"""The N > 1 path on CPU: two gloo ranks shard a frame list with no data-path collective; the only communication
is the barrier / MAX reduction bench.py uses for its timing.  (The per-frame work itself needs the MI355X; here
each rank runs the CPU oracle on its shard, which is what the sharding logic is independent of.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import graph_oracle as go
    from radargnn_amd import synthetic
    from radargnn_amd.frames import shard_range
    lo, hi = shard_range(n_frames, rank, world)
    edges = []
    for i in range(lo, hi):                                   # frames are independent: no exchange of data
        f = synthetic.nuscenes_frame(i)
        edges.append(go.radius_edges(f.X, 6.0).shape[0])
    dist.barrier()
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                  # bench.py: max-over-ranks of the elapsed time
    counts = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(counts)                                   # reporting only
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([lo, hi, int(counts.item()), float(t.item())] + edges))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_cover_all_frames_once(tmp_path):
    world, n_frames = 2, 7
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    from oracle import graph_oracle as go
    from radargnn_amd import synthetic
    got = {}
    for r in range(world):
        a = np.load(tmp_path / f"rank{r}.npy")
        lo, hi, total, tmax = int(a[0]), int(a[1]), int(a[2]), a[3]
        assert total == n_frames and abs(tmax - 0.2) < 1e-12
        for i, e in zip(range(lo, hi), a[4:]):
            assert i not in got
            got[i] = int(e)
    assert sorted(got) == list(range(n_frames))
    for i in (0, 6):
        assert got[i] == go.radius_edges(synthetic.nuscenes_frame(i).X, 6.0).shape[0]
