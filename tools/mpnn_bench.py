"""A/B of edge-kernel (rgnn_mpnn_aggregate) variants on the C2 graph (tools only).

    python tools/mpnn_bench.py [-r rounds] [-d 464] variant [variant ...]

A variant is  path/to/librgnn.so[:ENV=VALUE...]  (see tools/x3_bench.hip); the first one is the reference the others are
compared with bit for bit.  The graph (64 RadarScenes-shaped frames, radius 1 m, grid-cell visiting order, work-balanced
chunks) is built once with the product library; Q, W_e and the edge hidden activations are random.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radargnn_amd import _lib, frames as fr, ops, synthetic
from radargnn_amd.gnn.mpnn_layers import TargetCSR


def load(spec):
    path, *envs = spec.split(":")
    lib = C.CDLL(os.path.abspath(path), mode=os.RTLD_LOCAL if hasattr(os, "RTLD_LOCAL") else 0)
    for name in ("rgnn_mpnn_aggregate", "rgnn_mpnn_partition", "rgnn_mpnn_num_chunks"):
        res, args = _lib.SIGNATURES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return spec, lib, [e.split("=", 1) for e in envs]


def main():
    argv = sys.argv[1:]
    rounds, d, aggr = 7, 464, 0
    knn = 0
    specs = []
    while argv:
        a = argv.pop(0)
        if a == "-r":
            rounds = int(argv.pop(0))
        elif a == "-d":
            d = int(argv.pop(0))
        elif a == "-k":                                  # kNN graph with k neighbours instead of the radius graph (C4: -k 20)
            knn = int(argv.pop(0))
        else:
            specs.append(a)
    variants = [load(s) for s in specs]
    frames = [synthetic.radarscenes_frame(i) for i in range(64)]
    if knn:
        g = fr.build_graphs(fr.FrameBatch.from_frames(frames), fr.GraphSettings(algorithm="knn", k=knn))
        csr = TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order, all_sources=True)
    else:
        g = fr.build_graphs(fr.FrameBatch.from_frames(frames), fr.GraphSettings(algorithm="radius", r=1.0))
        csr = TargetCSR(g.edge_index, g.x.shape[0], order=g.cell_order, symmetric=True)
    n, e = g.x.shape[0], csr.num_edges
    torch.manual_seed(0)
    Q = torch.randn(n, d, device="cuda")
    We = torch.randn(d, 8, device="cuda") * 0.3
    ea = torch.randn(e, 8, device="cuda").relu_()
    stream = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: C.c_void_p(t.data_ptr())
    outs, times = [], [[] for _ in variants]

    def run(i, out):
        _, lib, envs = variants[i]
        for k, v in envs:
            os.environ[k] = v
            __import__("radargnn_amd").ops.reload_env()
        # every variant partitions for itself (the chunking constants may differ between libraries)
        nc = lib.rgnn_mpnn_num_chunks(n, e)
        chunks = run.chunks.setdefault(i, torch.empty(nc + 1 + 1024, dtype=torch.int32, device="cuda"))
        if i not in run.done:
            assert lib.rgnn_mpnn_partition(ptr(csr.rowptr), n, e, ptr(chunks), stream) == 0
            run.done.add(i)
        rc = lib.rgnn_mpnn_aggregate(None, 0, None, ptr(Q), d, ptr(We), 8, ptr(ea), 8, ptr(csr.rowptr), ptr(csr.src),
                                     ptr(csr.order), ptr(chunks), nc, n, d, aggr, ptr(out), d, stream)
        for k, _ in envs:
            os.environ.pop(k, None)
            __import__("radargnn_amd").ops.reload_env()
        assert rc == 0, rc

    run.chunks, run.done = {}, set()
    for i in range(len(variants)):
        out = torch.zeros(n, d, device="cuda")
        run(i, out)
        torch.cuda.synchronize()
        outs.append(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    scratch = torch.empty(n, d, device="cuda")
    for _ in range(rounds):
        for i in range(len(variants)):
            run(i, scratch)
            e0.record()
            for _ in range(5):
                run(i, scratch)
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 5)
    print(f"edge kernel: N={n} E={e} D={d}  gather volume {4 * e * d / 1e9:.2f} GB, rows with edges "
          f"{int((csr.rowptr[1:] > csr.rowptr[:-1]).sum())}")
    for i, (spec, _, _) in enumerate(variants):
        t = sorted(times[i])
        med = t[len(t) // 2]
        bad = int((outs[i] != outs[0]).sum()) if i else 0
        print(f"    {spec:60s} {med * 1e3:7.1f} us (min {t[0] * 1e3:7.1f})  {4 * e * d / med / 1e9:6.2f} TB/s gathered"
              + (f"  x{sorted(times[0])[len(t) // 2] / med:.3f} vs first | mismatches {bad}" if i else ""))


if __name__ == "__main__":
    main()
