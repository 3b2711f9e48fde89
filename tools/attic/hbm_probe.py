"""What this box's HBM delivers to simple streaming kernels (torch's own elementwise / reduction launches), as a yardstick for the
traffic-bound kernels of the step: read-only, write-only, copy (1 read : 1 write) and 2 reads : 1 write, on buffers far larger than
L2 + MALL.  GB/s = bytes the kernel must move / HIP-event time, best and median of 20."""
import torch

def timed(fn, n=20):
    ts = []
    for _ in range(3):
        fn()
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]

def main():
    for mb in (512, 2048):
        n = mb * (1 << 20) // 4
        x = torch.randn(n, device="cuda"); y = torch.empty_like(x); z = torch.randn(n, device="cuda")
        rows = [("read  (sum)", lambda: x.sum(), 4 * n), ("write (fill_)", lambda: y.fill_(1.0), 4 * n),
                ("copy  (copy_)", lambda: y.copy_(x), 8 * n), ("2r:1w (add out=)", lambda: torch.add(x, z, out=y), 12 * n),
                ("1r:1w in place (mul_)", lambda: x.mul_(1.0000001), 8 * n)]
        for name, fn, nbytes in rows:
            best, med = timed(fn)
            print(f"{mb:5d} MB buffers  {name:24s} best {nbytes / best / 1e9:7.0f} GB/s   median {nbytes / med / 1e9:7.0f} GB/s")

if __name__ == "__main__":
    main()
