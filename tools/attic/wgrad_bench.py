"""Weight-gradient kernels on the layer shapes of a C2 training step: rgnn_wgrad (bf16x3) against rgnn_linear_wgrad (fp32 MFMA).
    python tools/attic/wgrad_bench.py [variant librgnn.so ...]      (variants: tools/build_variant.sh; compared with the product library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from radargnn_amd import ops


def timed(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


def variant(path):
    import ctypes as C
    from radargnn_amd import _lib
    v = C.CDLL(os.path.abspath(path))
    for name in ("rgnn_wgrad", "rgnn_wgrad_slabs"):
        res, args = _lib.SIGNATURES[name]
        fn = getattr(v, name)
        fn.restype, fn.argtypes = res, args
    return v


class use:
    """ops.lib.rgnn_wgrad* taken from another library inside the block"""
    def __init__(self, v):
        self.v = v
    def __enter__(self):
        self.saved = (ops.lib.rgnn_wgrad, ops.lib.rgnn_wgrad_slabs)
        if self.v is not None:
            ops.lib.rgnn_wgrad, ops.lib.rgnn_wgrad_slabs = self.v.rgnn_wgrad, self.v.rgnn_wgrad_slabs
    def __exit__(self, *a):
        ops.lib.rgnn_wgrad, ops.lib.rgnn_wgrad_slabs = self.saved


def main():
    variants = [(p, variant(p)) for p in sys.argv[1:]]
    torch.manual_seed(0)
    m = 192000
    small = [("edge emb [16 x 8] E rows", 16, 8, 0, 0, 800000), ("edge emb [8 x 4] E rows", 8, 4, 0, 0, 800000), ("edge emb [4 x 2] E rows", 4, 2, 0, 0, 800000),
             ("node emb [32 x 5]", 32, 5, 0, 0, m), ("reg head [5 x 16]", 5, 16, 0, 0, m), ("node emb [64 x 32]", 64, 32, 0, 0, m),
             ("node emb [224 x 128]", 224, 128, 0, 0, m), ("cls head [6 x 64]", 6, 64, 0, 0, m), ("reg head [16 x 64]", 16, 64, 0, 0, m)]
    for name, n, k1, k2, rows, mm in small:
        g = torch.randn(mm, n, device="cuda"); a1 = torch.randn(mm, k1, device="cuda")
        t3 = timed(lambda: ops.linear_wgrad(g, a1, None, with_bias=True))
        print(f"{name:26s} rows {mm:6d}: {t3 * 1e3:7.1f} us  ({mm * (n + k1) * 4 / t3 / 1e9:6.2f} TB/s of rows streamed)", flush=True)
    for name, n, k1, k2, rows in [("update  [224 x 688]", 224, 224, 464, 105600), ("source  [464 x 224]", 464, 224, 0, 105600),
                                  ("iso     [224 x 224]", 224, 224, 0, 86400), ("L4 upd  [64 x 400]", 64, 128, 272, 105600),
                                  ("emb     [128 x 64]", 128, 64, 0, 0), ("emb     [32 x 5]", 32, 5, 0, 0), ("dense   [224 x 688]", 224, 224, 464, 0)]:
        g = torch.randn(m, n, device="cuda")
        a1 = torch.randn(m, k1, device="cuda")
        a2 = torch.randn(m, k2, device="cuda") if k2 else None
        lst = cnt = None
        mm = m
        if rows:
            perm = torch.randperm(m, device="cuda")[:rows].sort().values.int()
            lst = torch.zeros(m, dtype=torch.int32, device="cuda")
            lst[:rows] = perm
            cnt = torch.tensor([rows], dtype=torch.int64, device="cuda")
            mm = rows
        fl = 2.0 * mm * n * (k1 + k2)
        t3 = timed(lambda: ops.linear_wgrad(g, a1, a2, with_bias=True, row_index=lst, m_dev=cnt))
        line = f"{name:22s} rows {mm:6d}: bf16x3 {t3 * 1e3:7.1f} us {fl / t3 / 1e9:6.1f} TF"
        if not rows and k1 % 4 == 0:
            t1 = timed(lambda: ops.linear_wgrad_fp32(g, a1, a2))
            line += f" | fp32 MFMA {t1 * 1e3:7.1f} us {fl / t1 / 1e9:6.1f} TF"
        for path, v in variants:
            with use(v):
                tv = timed(lambda: ops.linear_wgrad(g, a1, a2, with_bias=True, row_index=lst, m_dev=cnt))
            line += f" | {os.path.basename(os.path.dirname(path))} {tv * 1e3:7.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
