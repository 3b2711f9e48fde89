"""Host-side cost of one eager step of the C2 hot path (python + ctypes + allocator), to keep the launch queue ahead of the GPU."""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from radargnn_amd import frames as fr, synthetic

model = bench.c2_model().cuda()
settings = fr.GraphSettings(algorithm="radius", k=0, r=1.0)
hot = fr.HotPath(model, settings, use_hip_graphs=False)
batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(64)])
for _ in range(5):
    hot(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    hot(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    hot(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
