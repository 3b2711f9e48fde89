"""Device-side collation on the C2 batch shape: 64 graphs of 3000 nodes, ~12.5 k edges each, drawn in a
random order from a resident store of 512 graphs.  Prints one JSON line: algorithmic bytes (read + write of every
attribute) / time against the 8 TB/s HBM peak, and the host path of the reference (torch.cat on the CPU + H2D copy) on
the same graphs beside it."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radargnn_amd import data as D  # noqa: E402


def main():
    g = torch.Generator().manual_seed(0)
    graphs = []
    for i in range(512):
        n, e = 3000, 12000 + int(torch.randint(0, 1000, (1,), generator=g))
        graphs.append(D.Data(x=torch.randn(n, 5, generator=g), edge_index=torch.randint(0, n, (2, e), generator=g),
                             edge_attr=torch.randn(e, 2, generator=g), y=torch.randn(n, 6, generator=g),
                             pos=torch.randn(n, 2, generator=g), vel=torch.randn(n, 2, generator=g)))
    store = D.GraphStore(graphs)
    ids = torch.randperm(512, generator=g)[:64].numpy()
    b = store.collate(ids)
    torch.cuda.synchronize()
    nbytes = 2 * sum(getattr(b, k).numel() * getattr(b, k).element_size() for k in store.keys) + b.batch.numel() * 8
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        b = store.collate(ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tabs = store.collate(ids)                                  # device time alone: events around the launches of one call
    ev0.record()
    for _ in range(reps):
        store.collate(ids)
    ev1.record(); torch.cuda.synchronize()
    dev_ms = ev0.elapsed_time(ev1) / reps
    # the reference's path: collate on the host, then copy to the device (inference.py:57)
    sel = [graphs[i] for i in ids]
    t0 = time.perf_counter()
    for _ in range(5):
        off = np.concatenate(([0], np.cumsum([s.num_nodes for s in sel])))
        host = {k: torch.cat([s[k] for s in sel], 0) for k in ("x", "edge_attr", "y", "pos", "vel")}
        host["edge_index"] = torch.cat([s.edge_index + int(o) for s, o in zip(sel, off)], 1)
        host["batch"] = torch.repeat_interleave(torch.arange(64), torch.tensor([s.num_nodes for s in sel]))
        dev = {k: v.cuda() for k, v in host.items()}
        torch.cuda.synchronize()
    host_dt = (time.perf_counter() - t0) / 5
    print(json.dumps({"what": "collate 64 of 512 resident graphs (C2 shape) into one Batch", "nodes": int(b.x.shape[0]),
                      "edges": int(b.edge_index.shape[1]), "algorithmic_bytes": int(nbytes), "ms_per_batch_wall": dt * 1e3,
                      "ms_per_batch_device_stream": dev_ms, "achieved_GBps_wall": nbytes / dt / 1e9, "peak_GBps": 8000.0,
                      "frac_wall": nbytes / dt / 8e12, "host_collate_plus_h2d_ms": host_dt * 1e3,
                      "speedup_vs_host_path": host_dt / dt, "resident_store_bytes": store.nbytes()}))


if __name__ == "__main__":
    main()
