"""Which ATen launches one HotPath step still makes, and from where: a TorchDispatchMode that logs every op that is not a view /
allocation together with the innermost radargnn_amd frames of the Python stack.  (The step's own kernels go through ctypes and do
not show up here: what is listed is what is NOT a hand-written launch.)"""
import sys, os, traceback, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from radargnn_amd import frames as fr, synthetic

SKIP = ("empty", "view", "as_strided", "slice", "select", "detach", "reshape", "t.", "transpose", "_unsafe_view", "alias", "unsqueeze",
        "expand", "resize_", "_local_scalar_dense", "lift_fresh", "squeeze", "permute", "narrow", "split", "unbind", "new_empty", "numel",
        "sym_", "stride", "is_", "_to_copy")

class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not name.startswith(SKIP):
            st = [f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in traceback.extract_stack() if "radargnn_amd" in f.filename]
            self.seen[(name, " < ".join(reversed(st[-3:])))] += 1
        return func(*args, **(kwargs or {}))

def main():
    train = "--train" in sys.argv
    model = bench.c2_model().cuda()
    batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(8)])
    hot = fr.HotPath(model, bench.c2_settings())
    for _ in range(3):
        hot(batch)
    torch.cuda.synchronize()
    with Log() as log:
        hot(batch)
        torch.cuda.synchronize()
    for (name, where), n in sorted(log.seen.items(), key=lambda kv: -kv[1]):
        print(f"{n:3d} x {name:28s} {where}")

if __name__ == "__main__":
    main()
