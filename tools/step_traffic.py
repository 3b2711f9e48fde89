"""profiles/<tag>_step_traffic.json from <tag>_pmc_hbm_traffic.json (tools/profile_round.sh -> tools/summarize_profiles.py): HBM bytes one C2
step moves, by the counters: sum over kernels of dispatches x mean bytes per launch, divided by the steps of the profiled command (the
dispatch count of k_mpnn_win / 4: four conv layers per step; warm-up, probe and timed steps alike).
    python tools/step_traffic.py gpurun_out/r05_summary/r05_pmc_hbm_traffic.json profiles/r05_step_traffic.json"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
win = next(v for k, v in d.items() if "k_mpnn_win" in k)
steps = win["dispatches"] / 4
total = sum(v["dispatches"] * v["hbm_bytes_per_launch"] for v in d.values())
top = sorted(d.items(), key=lambda kv: -kv[1]["dispatches"] * kv[1]["hbm_bytes_per_launch"])[:12]
out = {"source": f"{src.split('/')[-1]} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 10 --warmup 3 "
                 "--no-cpu-baseline --no-other-configs --no-pcie`, FETCH doubled on gfx950; RGNN_NO_PLAN_SIDE=1)",
       "steps_in_the_profiled_command": steps, "hbm_bytes_per_step": total / steps,
       "mb_per_step_by_kernel_top12": {k: round(v["dispatches"] * v["hbm_bytes_per_launch"] / steps / 1e6, 1) for k, v in top}}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
