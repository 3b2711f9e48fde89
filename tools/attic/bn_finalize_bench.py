"""Time rgnn_batchnorm_finalize_bound on the C2 shape (two row-subset launches' panels, 224 channels):
    python tools/attic/bn_finalize_bench.py        (RGNN_BN_FIN_ABL = 1 | 2 | 4 | 7: timing experiments of the vectorised kernel, -1: scalar kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radargnn_amd import ops

n, rows_a, rows_b = 224, 133516, 58484
panels = ops.stat_panels(192000)
st = torch.randn(2 * panels, ops.STAT_ROWS, n, device="cuda")
st[:, 0, :] = 128.0
ca, cb = torch.tensor([rows_a], device="cuda"), torch.tensor([rows_b], device="cuda")
gamma, beta = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda")
rm, rv, nb = torch.zeros(n, device="cuda"), torch.ones(n, device="cuda"), torch.zeros((), dtype=torch.int64, device="cuda")
parts = ops.StatParts([(st[:panels], ca), (st[panels:], cb)])
big = torch.empty(64 << 20, device="cuda")
for abl in os.environ.get("ABLS", "0,1,2,4,7,-1").split(","):
    os.environ["RGNN_BN_FIN_ABL"] = abl
    __import__("radargnn_amd").ops.reload_env()
    ts = []
    for it in range(30):
        big.fill_(1.0)                                     # (the panels come from HBM / MALL, as in the step, not from L2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.batchnorm_finalize(parts, 192000, n, gamma, beta, rm, rv, nb, True, 0.1, 1e-5)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"ABL {abl:>3}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f} us")
