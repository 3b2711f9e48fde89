"""Do two row-subset dense launches overlap when issued on two streams?  L1 shapes of the C2 workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radargnn_amd import ops

torch.manual_seed(0)
n, c, d, co = 192000, 224, 464, 224
n_ne = 105600
x = torch.randn(n, c).cuda(); M = torch.randn(n, d).cuda()
wm = torch.randn(co, c + d).cuda() / 26; wi = torch.randn(co, c).cuda() / 15; b = torch.randn(co).cuda()
perm = torch.randperm(n)
lst_ne = torch.zeros(n, dtype=torch.int32); lst_ne[:n_ne] = perm[:n_ne].sort().values.int()
lst_e = torch.zeros(n, dtype=torch.int32); lst_e[:n - n_ne] = perm[n_ne:].sort().values.int()
lst_ne, lst_e = lst_ne.cuda(), lst_e.cuda()
cnt_ne = torch.tensor([n_ne]).cuda(); cnt_e = torch.tensor([n - n_ne]).cuda()
h = torch.empty(n, co).cuda()
side = torch.cuda.Stream()


def main_launch():
    ops.linear(x, wm, b, a2=M, out=h, row_index=lst_ne, m_dev=cnt_ne)


def iso_launch():
    ops.linear(x, wi, b, out=h, row_index=lst_e, m_dev=cnt_e)


def seq():
    main_launch(); iso_launch()


def par():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    main_launch()
    with torch.cuda.stream(side):
        iso_launch()
    cur.wait_stream(side)


def par_first():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        iso_launch()
    main_launch()
    cur.wait_stream(side)


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("main alone      %.1f us" % timeit(main_launch))
print("iso alone       %.1f us" % timeit(iso_launch))
print("sequential      %.1f us" % timeit(seq))
print("two streams     %.1f us" % timeit(par))
print("iso first       %.1f us" % timeit(par_first))

# with an upstream kernel both launches depend on (the edge kernel in the model): who starts first is decided by the queues
pre_a = torch.randn(150000, 688).cuda(); pre_w = torch.randn(224, 688).cuda()


def pre():
    ops.linear(pre_a, pre_w, None)


def dep_seq():
    pre(); main_launch(); iso_launch()


def dep_main_on_cur():
    cur = torch.cuda.current_stream()
    pre()
    side.wait_stream(cur)
    main_launch()
    with torch.cuda.stream(side):
        iso_launch()
    cur.wait_stream(side)


def dep_iso_on_cur():
    cur = torch.cuda.current_stream()
    pre()
    side.wait_stream(cur)
    iso_launch()
    with torch.cuda.stream(side):
        main_launch()
    cur.wait_stream(side)


t_pre = timeit(pre)
print("pre alone                      %.1f us" % t_pre)
print("pre + sequential               %.1f us (pair: %.1f)" % (timeit(dep_seq), timeit(dep_seq) - t_pre))
print("pre + main on cur, iso on side %.1f us (pair: %.1f)" % (timeit(dep_main_on_cur), timeit(dep_main_on_cur) - t_pre))
print("pre + iso on cur, main on side %.1f us (pair: %.1f)" % (timeit(dep_iso_on_cur), timeit(dep_iso_on_cur) - t_pre))
