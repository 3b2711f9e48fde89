"""Which ATen launches one C2 TRAINING step makes, and from where: TorchDispatchMode log (see prof_aten_ops.py) with the backward
pass run on the calling thread so that the ops of the custom Functions' backward methods are seen too."""
import sys, os, traceback, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from radargnn_amd import frames as fr, synthetic
from radargnn_amd.gnn.losses import detection_loss

SKIP = ("empty", "view", "as_strided", "slice.", "select.", "detach", "reshape", "t.", "transpose", "_unsafe_view", "alias", "unsqueeze",
        "expand", "resize_", "_local_scalar_dense", "lift_fresh", "squeeze", "permute", "narrow", "split", "unbind", "new_empty", "numel",
        "sym_", "stride", "is_", "_to_copy", "result_type")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not name.startswith(SKIP):
            st = [f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in traceback.extract_stack()
                  if "radargnn_amd" in f.filename or "optim" in f.filename]
            big = max([a.numel() for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)] + [0])
            size = "HUGE(>=1e6)" if big >= 1_000_000 else ("mid" if big >= 20_000 else "small")
            self.seen[(name + " " + size, " < ".join(reversed(st[-2:])))] += 1
        return func(*args, **(kwargs or {}))


def main():
    model = bench.c2_model().cuda()
    batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(16)])
    g = fr.build_graphs(batch, bench.c2_settings())
    n = g.x.shape[0]
    y = torch.cat((torch.randint(0, 6, (n, 1), device="cuda").float(), torch.randn(n, 5, device="cuda")), 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, ei, ea = g.x, g.edge_index, g.edge_attr

    def step():
        opt.zero_grad(); x.requires_grad_(); ea.requires_grad_()
        c, bb = model(x, ei, ea)
        loss, _, _ = detection_loss(c, bb, y, 5, [1.0] * 5 + [0.3])
        loss.backward(); opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with torch.autograd.set_multithreading_enabled(False), Log() as log:
        step()
        torch.cuda.synchronize()
    tot = sum(log.seen.values())
    print(f"{tot} logged ATen ops in one training step")
    for (name, where), k in sorted(log.seen.items(), key=lambda kv: -kv[1])[:70]:
        print(f"{k:3d} x {name:42s} {where}")


if __name__ == "__main__":
    main()
