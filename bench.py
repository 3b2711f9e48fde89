#!/usr/bin/env python
"""Hot-path benchmark: radar frames/s of (graph-build + features + DetNetBasic forward) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch resident in HBM.  Workload = BASELINE.json configs[1] ("C2"):
per GPU a batch of 64 RadarScenes-shaped synthetic frames (~3000 points each), radius graph r = 1.0,
translation-invariant features, 4-layer MPNNConv [224,224,128,64] with the shipped embedding MLPs and both heads,
module in training mode (the reference never calls .eval(): batch statistics in every BatchNorm).  Frames are
independent -> weak scaling: every rank owns its own 64-frame batch, no collective in the data path; the only
communication is the barrier / max-over-ranks of the timing.

Rank 0 prints ONE JSON line (see MEASUREMENTS.md "Measurement" for the definitions of `roofline` and `cpu_baseline`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FRAMES_PER_GPU = 64
PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: ~2.5 PF dense bf16 (v_mfma_f32_32x32x16_bf16)
PEAK_HBM_GBS = 8000.0
PEAK_L2_GBS = 34500.0              # same guide, section L2: ~34.5 TB/s aggregate


class EventProfiler:
    """HIP-event pairs recorded INSIDE librgnn immediately around the launches of the two dominant kernels
    (rgnn_profile_next_launch), on torch's current stream = the stream librgnn launches on."""

    def __init__(self, rows_with_edges=0, symmetric=False):
        self.records = []
        self.enabled = False
        self.rows_with_edges = rows_with_edges      # nodes with incoming edges in the batch (edge-kernel compulsory bytes)
        self.symmetric = symmetric

    def begin(self, kind):
        if not self.enabled:
            return None
        from radargnn_amd import ops
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record(); e.record()                       # instantiate the underlying hipEvent_t handles
        ops.arm_profile_events(s, e)
        return (kind, s, e)

    def end(self, tok, **work):
        self.records.append((tok[0], tok[1], tok[2], work))

    def summary(self):
        out = {}
        for kind, s, e, work in self.records:
            ms = s.elapsed_time(e)
            d = out.setdefault(kind, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "big_ms": 0.0, "big_flops": 0.0,
                                      "big_launches": 0, "x3_ms": 0.0, "x3_flops": 0.0, "x3_launches": 0,
                                      "x3_exec_flops": 0.0, "f16_launches": 0})
            d["launches"] += 1
            d["ms"] += ms
            if kind == "linear":
                m_rows = work["m"]
                if torch.is_tensor(m_rows):                  # row-subset launch: the row count is a device scalar
                    m_rows = int(m_rows.item())
                fl = 2.0 * m_rows * work["n"] * work["k"]
                d["flops"] += fl
                if work["n"] > 64:                     # the wide tile instances: the dominant kernel symbols
                    d["big_ms"] += ms; d["big_flops"] += fl; d["big_launches"] += 1
                    if work.get("x3"):                 # ... of which: launches on the bf16x3 kernel
                        # matrix-pipe products per fp32 product: 3 in the f16x2 form (two f16 terms per operand), 6 in bf16x3
                        prod = 3.0 if work.get("f16") else 6.0
                        d["x3_ms"] += ms; d["x3_flops"] += fl; d["x3_launches"] += 1
                        d["x3_exec_flops"] += prod * fl
                        # bytes the launch must move once: its activation rows in, its output rows out, the weight (two f16
                        # or three bf16 planes are 4 or 6 bytes per element; counted as fp32) -- DESIGN.md section 5
                        d["x3_bytes"] = d.get("x3_bytes", 0.0) + 4.0 * m_rows * (work["k"] + work["n"]) + 4.0 * work["n"] * work["k"]
                        d["f16_launches"] += 1 if work.get("f16") else 0
                        if ms > 0 and prod * fl / ms > d.get("x3_best", 0.0):
                            d["x3_best"] = prod * fl / ms     # executed flop per ms of the launch that ran fastest
            elif kind == "mpnn_aggregate":
                # L2-level gather volume: one D-wide row of Q per edge.  Compulsory HBM bytes: the Q rows that exist (sources
                # = the nodes with edges in a symmetric graph, all nodes otherwise) once, the edge stream (attributes + source
                # index), the CSR slice (row pointer + visiting order), the rows written (targets with edges when the caller
                # skips the others).
                rows_q = self.rows_with_edges if self.symmetric else work["n"]
                rows_out = self.rows_with_edges if self.symmetric else work["n"]
                d["gather_bytes"] = d.get("gather_bytes", 0.0) + 4.0 * work["e"] * work["d"]
                d["win_launches"] = d.get("win_launches", 0) + (1 if work.get("win") else 0)
                d["bytes"] += (4.0 * work["d"] * rows_q + work["e"] * (4.0 * work["de"] + 4.0) + 8.0 * work["n"]
                               + 4.0 * work["d"] * rows_out)
                d["flops"] += work["e"] * (2.0 * work["de"] * work["d"] + work["d"])
        return out


def c2_model(seed=0):
    from radargnn_amd import gnn
    cfg = gnn.GNNArchitectureConfig(
        node_feature_dimension=5, edge_feature_dimension=2, conv_layer_dimensions=[224, 224, 128, 64],
        classification_head_layer_dimensions=[6], regression_head_layer_dimensions=[16, 5],
        initial_node_feature_embedding=True, initial_edge_feature_embedding=True,
        node_feature_embedding_layer_dimensions=[32, 64, 128, 224], edge_feature_embedding_layer_dimensions=[4, 8, 16],
        conv_layer_type="MPNNConv", batch_norm_in_mlps=False)
    torch.manual_seed(seed)
    return gnn.DetNetBasic(cfg)


def c2_settings():
    from radargnn_amd import frames as fr
    return fr.GraphSettings(algorithm="radius", k=0, r=1.0)


def shipped_model(dims, k_classes, node_dim=5, edge_dim=2, seed=0):
    """configurations/configuration_radarscenes.yml:17-41 (nuScenes: 11 classes): embeddings [32,64,128,224] / [4,8,16]."""
    from radargnn_amd import gnn
    torch.manual_seed(seed)
    return gnn.DetNetBasic(gnn.GNNArchitectureConfig(node_dim, edge_dim, dims, [k_classes], [16, 5], True, True,
                                                     [32, 64, 128, 224], [4, 8, 16], "MPNNConv", False))


def cpu_baseline(model, settings, n_frames=4, warmups=3, reps=10):
    """Reference-shaped CPU path (oracle/reference_shaped.py) on a bounded sample of the same workload, SURVEY 8(d)
    protocol: 3 warm-ups, then >= 10 timed repetitions of graph-build + forward, median.  The graph stage is one Python
    process like the reference's per-frame code; the forward runs on the torch thread count that a short probe finds
    fastest (all 256 hardware threads of the box are ~70x SLOWER than 16 for these small eager ops)."""
    import statistics
    from oracle import reference_shaped
    from radargnn_amd import synthetic
    frames = [synthetic.radarscenes_frame(i) for i in range(n_frames)]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    args = (frames, settings.algorithm, settings.k, settings.r, list(settings.node_features), list(settings.edge_features),
            settings.edge_mode, sd)
    ncpu = os.cpu_count() or 1
    probe = {}
    for thr in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu)}):
        torch.set_num_threads(thr)
        probe[thr] = reference_shaped.time_forward_only(*args)
    thr = min(probe, key=probe.get)
    torch.set_num_threads(thr)
    runs = [reference_shaped.time_hot_path(*args) for _ in range(warmups + reps)][warmups:]
    total = statistics.median(r["graph_s"] + r["forward_s"] for r in runs)
    graph_s = statistics.median(r["graph_s"] for r in runs)
    fwd_s = statistics.median(r["forward_s"] for r in runs)
    vec = [reference_shaped.time_vectorised(*args) for _ in range(1 + 5)][1:]
    vec_total = statistics.median(v["graph_s"] + v["forward_s"] for v in vec)
    # parity of the SAME sample, measured in this run: the HIP path's logits / boxes on these frames against the float64 oracle on
    # the oracle's own graphs (the oracle is the checker here, as everywhere)
    live = None
    try:
        import numpy as np
        from oracle import gnn_oracle, graph_oracle
        from radargnn_amd import frames as fr
        cls, bb, g = fr.HotPath(model, settings)(fr.FrameBatch.from_frames(frames))
        ref = graph_oracle.collate([graph_oracle.build_frame_graph(f.X, f.V, f.rcs, f.timestamp, settings.algorithm, settings.k, settings.r,
                                                                  list(settings.node_features), list(settings.edge_features),
                                                                  settings.edge_mode) for f in frames])
        same_graph = bool(np.array_equal(g.edge_index.cpu().numpy(), ref["edge_index"]) and np.array_equal(g.x.cpu().numpy(), ref["x"]))
        c64, b64 = gnn_oracle.det_net_basic(torch.from_numpy(ref["x"]), torch.from_numpy(ref["edge_index"]),
                                            torch.from_numpy(ref["edge_attr"]), sd, dtype=torch.float64)
        live = {"frames": n_frames, "topology_and_node_features_bit_equal": same_graph,
                "logits": ((cls.double().cpu() - c64).abs().max() / c64.abs().max()).item(),
                "boxes": ((bb.double().cpu() - b64).abs().max() / b64.abs().max()).item()}
    except Exception as exc:                                  # (a reported extra: never the reason a bench line is missing)
        live = {"error": repr(exc)[:200]}
    return {
        "parity_on_this_sample": live,
        "value": n_frames / total, "unit": "frames/s", "cores": thr, "kind": "port",
        "sample": f"{n_frames} of the {FRAMES_PER_GPU} frames as one batch, {warmups} warm-ups + {reps} repetitions, median; "
                  f"reference-shaped path: sklearn KD-tree + dense adjacency + networkx degree + one Python iteration per "
                  f"edge (1 process, like the reference's per-frame code) = {graph_s:.2f}s; eager torch gather/cat/Linear/"
                  f"scatter forward on {thr} threads (fastest of 8/16/32 in a probe) = {fwd_s:.2f}s",
        "vectorised_value": n_frames / vec_total,
        "vectorised_note": "numpy-vectorised oracle (no dense adjacency, no per-edge Python) + the same forward, median of 5",
    }


def pcie_inclusive(hot, frames_list, steps):
    """frames/s with the boundary's host buffers on BOTH sides: every step's frames come from host memory and its logits / boxes
    go back to it (what postprocessor/inference.py:57,65-68 does per frame), through frames.FrameStreamer -- pinned staging
    filled by a loader thread, H2D on a copy stream under the previous batch's kernels, D2H on a third stream.  Reported
    beside `value` (which is resident-input throughput by contract), never as it."""
    from radargnn_amd import frames as fr
    streamer = fr.FrameStreamer(hot)
    # Steady state: the first batches pin the staging rings and grow the caching allocator's pools (a first-use hipMalloc /
    # hipHostMalloc stalls the device for tens of ms -- r04 saw one 90-ms stall land in batch 3 or 4 of some runs, which a 27-ms
    # timed window turned into 6 k frames/s instead of 25 k): several turns of the rings before the clock starts, and a window long
    # enough (>= 200 batches, ~0.45 s) that one such stall cannot decide the number; the longest interval is reported beside it
    warm = max(4 * streamer.slots, 24)

    def batches():
        for _ in range(warm + steps):
            yield frames_list

    stamps = []
    for _cls, _bb in streamer.run(batches()):
        stamps.append(time.perf_counter())
    timed = stamps[warm - 1:]                              # `steps` intervals behind the warm-up
    gaps = sorted(b - a for a, b in zip(timed[:-1], timed[1:]))
    return {"value": len(frames_list) * steps / (timed[-1] - timed[0]),
            "median_interval_value": len(frames_list) / gaps[len(gaps) // 2], "batches": steps,
            "longest_interval_ms": gaps[-1] * 1e3, "slots": streamer.slots, "results_behind": streamer.behind}


def _read_counter_file(path, counter, per):
    """rocprofv3's <prefix>_counter_collection.csv -> per[kernel][counter] = [value per dispatch] (kernel = its short name)."""
    import csv
    import re
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
            key = m.group(1) if m else r["Kernel_Name"][:60]
            per.setdefault(key, {}).setdefault(counter, []).append(float(r["Counter_Value"]))


def _summarise_counters(per):
    """per[kernel] = {"FETCH_SIZE": [KB per dispatch], "WRITE_SIZE": [...]} -> LIVE_PMC (bytes per launch of the dominant dense kernel and
    of the edge kernel, bytes per step); returns None, or why nothing could be derived."""
    import re
    rows = {}
    for k, v in per.items():
        f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
        if f:
            rows[k] = (len(f), (2.0 * sum(f) / len(f) + (sum(w) / len(w) if w else 0.0)) * 1024.0)

    def wide(k):
        m = re.match(r"k_linear_dma<(\d+), (true|false), (\d+)", k)
        return m is not None and int(m.group(1)) * 32 > 64 and int(m.group(3)) >= 1

    def mean(sel):
        n = sum(rows[k][0] for k in sel)
        return sum(rows[k][0] * rows[k][1] for k in sel) / n if n else None
    win = [k for k in rows if k.startswith("k_mpnn_win")] or [k for k in rows if k.startswith("k_mpnn_max")]
    if not win:
        return "no edge kernel in the counter file"
    steps = sum(rows[k][0] for k in win) / 4.0                       # four conv layers per step (warm-up, probe and timed steps alike)
    how = ("measured in THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel trace only) of a child "
           "`bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-pcie` started by this process; FETCH doubled on gfx950")
    LIVE_PMC["pmc_linear_summary.json"] = {"hbm_bytes_per_launch": mean([k for k in rows if wide(k)]), "live": how}
    LIVE_PMC["pmc_mpnn_summary.json"] = {"hbm_bytes_per_launch": mean(win), "edge_kernel": "k_mpnn_win" if win[0].startswith("k_mpnn_win") else "k_mpnn_max",
                                         "live": how}
    LIVE_PMC["r05_step_traffic.json"] = {"hbm_bytes_per_step": sum(n * b for n, b in rows.values()) / steps, "source": how,
                                         "steps_in_the_profiled_command": steps}
    return None


LIVE_PMC = {}        # name of a committed summary -> the same quantities measured in THIS run (live_traffic)


def live_traffic(timeout_s=120):
    """HBM traffic by the counters, measured in this run: the process starts `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
    passes, kernel trace only: MI355X_MICROARCH.md, HBM section) on a short child run of this very command (4 steps, no CPU baseline, no
    other configurations, no streaming) and reads the per-dispatch counters: mean bytes per launch of the dominant dense kernel
    (k_linear_dma<TN >= 3, ., FMT >= 1>: the launch set the event timing uses) and of the edge kernel, and the bytes of a whole step
    (FETCH doubled on gfx950).  Whatever goes wrong -- no rocprofv3, a pass that does not finish in time -- leaves LIVE_PMC empty and the
    line falls back to the committed summaries, labelled as such.  (the driver could not verify a number read from a file.)"""
    import csv
    import glob
    import re
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return "rocprofv3 not on PATH"
    root = tempfile.mkdtemp(prefix="rgnn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", RGNN_NO_PLAN_SIDE="1")      # (every kernel alone on the device: the counters mean what they say)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-other-configs",
             "--no-pcie", "--no-live-traffic"]
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(root, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "t", "--"] + child
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)               # (the group this call started: rocprofv3 and its child)
                proc.wait()
                return f"the {counter} pass did not finish within {timeout_s} s"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return f"the {counter} pass left no counter file (exit code {proc.returncode})"
            _read_counter_file(files[0], counter, per)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return _summarise_counters(per)


def _pmc_summary(name):
    if name in LIVE_PMC:
        return LIVE_PMC[name]
    path = os.path.join(REPO, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def survey_compulsory_bytes(n, e, conv_dims=(224, 224, 128, 64), c0=224, de=16, dn=5, de_raw=2):
    """bytes a FULLY FUSED step must move -- graph stage 48 N in, 16 E + 4 De E + 4 Dn N out; conv layer l:
    read 4 N C + 4 E De + 16 E, write 4 N Co (weights O(1))."""
    total = 48 * n + 16 * e + 4 * de_raw * e + 4 * dn * n
    c = c0
    for co in conv_dims:
        total += 4 * n * c + 4 * e * de + 16 * e + 4 * n * co
        c = co
    return float(total)


def step_traffic(n, e):
    """HBM bytes one C2 step moves by the counters (live_traffic: two --pmc passes of a short child run started by this process, behind
    the timed region; else the committed summary of the same passes) over SURVEY 8(d)'s fully fused compulsory bytes."""
    t = _pmc_summary("r05_step_traffic.json")
    comp = survey_compulsory_bytes(n, e)
    if not t:
        return {"survey_compulsory_bytes_per_step": comp, "hbm_bytes_per_step": None, "bytes_over_survey_compulsory": None}
    return {"survey_compulsory_bytes_per_step": comp, "hbm_bytes_per_step": t["hbm_bytes_per_step"],
            "bytes_over_survey_compulsory": t["hbm_bytes_per_step"] / comp, "source": t["source"]}


def parity_margin():
    """Worst norm-wise error (max|a - b| / max|b| per output tensor, float64 oracle) over the oracle comparisons at the TIMED sizes
    (tests/test_gpu_parity_timed_sizes.py: full C2 / C3 / C4 / C5 batches, written by the last `pytest -m gpu` run on a GPU box and
    committed), as a fraction of the 1e-5 bar."""
    t = _pmc_summary("r05_parity_margins.json")
    if not t:
        return None
    rows = {k: max(v.values()) for k, v in t.items() if "hoisted f64" not in k}
    worst = max(rows, key=rows.get)
    return {"worst_error": rows[worst], "worst_case": worst, "margin": rows[worst] / 1e-5, "bar": 1e-5, "cases": len(rows),
            "source": "profiles/r05_parity_margins.json (tests/test_gpu_parity_timed_sizes.py on an MI355X; not re-measured in this run)"}


def rooflines(summ, steps, with_pmc=True):
    """`roofline` (dense layers on the bf16 matrix pipe) and `roofline_gather` (edge kernel) from the event records."""
    lin = summ.get("linear", {})
    agg = summ.get("mpnn_aggregate", {})
    roofline = gather = None
    if lin.get("big_launches"):
        achieved = lin["big_flops"] / (lin["big_ms"] * 1e-3) / 1e12
        pmc = _pmc_summary("pmc_linear_summary.json") if with_pmc else None
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        if lin.get("x3_launches"):
            # every fp32 product is executed as THREE f16 MFMA products (f16x2 form: two f16 terms per operand after an exact
            # power-of-two pre-scale) or SIX bf16 products (bf16x3 form; launches whose operands carry no bound), fp32
            # accumulate: the roofline is the EXECUTED rate against the dense 16-bit MFMA peak (f16 and bf16 run at one rate)
            eq = lin["x3_flops"] / (lin["x3_ms"] * 1e-3) / 1e12
            ex = lin["x3_exec_flops"] / (lin["x3_ms"] * 1e-3) / 1e12
            # Which roof binds these launches is decided by their arithmetic intensity, executed flops per algorithmic byte,
            # against the machine balance (2 500 TFLOP/s / 8 TB/s = 312 flop/B): the bf16x3 form (6 products, r02) sat above
            # it, the f16x2 form (3 products: 3 * 2 K N / (4 (K + N)) = 226 flop/B at K, N = 224, 464 or 688, 224) sits below --
            # by the roofline model itself the dense layers are now bound by HBM, and that is the roof `frac` is taken against.
            # The matrix-pipe view of the same launches stays in `mfma` (what r01 / r02 reported as `roofline`).
            nb = lin.get("x3_bytes", 0.0)
            intensity = lin["x3_exec_flops"] / nb if nb else None
            balance = PEAK_BF16_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            gbs = nb / (lin["x3_ms"] * 1e-3) / 1e9
            hbm_bound = intensity is not None and intensity < balance
            roofline = {"bound": "hbm" if hbm_bound else "mfma",
                        "kernel": "k_linear_dma<TN,..,FMT> dense layer on the 16-bit matrix pipe (LDS-DMA staged; fp32 operands as "
                                  "2 f16 terms / 3 MFMA products per fp32 product where the operands carry a bound, else 3 bf16 "
                                  "terms / 6 products; fp32 accumulate); mean over ALL its launches with N > 64, the row-subset "
                                  "launches with their partial tile rounds included",
                        "achieved": gbs if hbm_bound else ex, "peak": PEAK_HBM_GBS if hbm_bound else PEAK_BF16_MFMA_TFLOPS,
                        "unit": "GB/s" if hbm_bound else "TFLOP/s",
                        "frac": gbs / PEAK_HBM_GBS if hbm_bound else ex / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic,
                        "traffic_source": (pmc.get("live") or "profiles/pmc_linear_summary.json (rocprofv3 --pmc passes of an earlier run of this "
                                           "command; not re-measured in this run)") if traffic else None,
                        "algorithmic_bytes_per_launch": nb / lin["x3_launches"],
                        "traffic_over_algorithmic": (traffic * lin["x3_launches"] / nb) if (traffic and nb) else None,
                        "traffic_gbs": (traffic / (lin["x3_ms"] / lin["x3_launches"] * 1e-3) / 1e9) if traffic else None,
                        "arithmetic_intensity_flop_per_byte": intensity, "machine_balance_flop_per_byte": balance,
                        "mfma": {"achieved": ex, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ex / PEAK_BF16_MFMA_TFLOPS},
                        "executed_products_per_fp32_product": lin["x3_exec_flops"] / lin["x3_flops"],
                        "f16x2_launches_per_step": lin["f16_launches"] / steps,
                        "fp32_equivalent_tflops": eq, "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS,
                        "fp32_equivalent_over_fp32_peak": eq / PEAK_FP32_MFMA_TFLOPS,
                        "best_launch_frac": lin.get("x3_best", 0.0) * 1e3 / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                        "measured": "HIP events around each launch, instrumented eager pass over the same steps",
                        "launches_per_step": lin["x3_launches"] / steps,
                        "avg_launch_ms": lin["x3_ms"] / lin["x3_launches"],
                        "flops_per_launch": lin["x3_flops"] / lin["x3_launches"],
                        "share_of_step_ms": lin["ms"] / steps}
        else:
            roofline = {"bound": "mfma", "kernel": "k_linear<...> fp32 MFMA dense layer (all tile instances with N > 64)",
                        "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                        "measured": "HIP events around each launch, instrumented eager pass over the same steps",
                        "launches_per_step": lin["big_launches"] / steps,
                        "avg_launch_ms": lin["big_ms"] / lin["big_launches"],
                        "flops_per_launch": lin["big_flops"] / lin["big_launches"],
                        "share_of_step_ms": lin["ms"] / steps}
    if agg.get("launches"):
        sec = agg["ms"] * 1e-3
        n = agg["launches"]
        pmc = _pmc_summary("pmc_mpnn_summary.json") if with_pmc else None
        win = agg.get("win_launches", 0)
        if pmc and pmc.get("edge_kernel", "k_mpnn_max") != ("k_mpnn_win" if win * 2 > n else "k_mpnn_max"):
            pmc = None                                   # (counters of the OTHER edge kernel: not this run's traffic)
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        comp = agg["bytes"] / sec / 1e9
        l2 = agg.get("gather_bytes", 0.0) / sec / 1e9
        hbm = (traffic * n / sec / 1e9) if traffic else None
        # three different questions, three numbers (r01 divided the L2-level gather volume by the HBM peak):
        #   achieved / frac          counter HBM bytes per launch / launch time vs 8 TB/s (falls back to the compulsory bytes)
        #   compulsory_*             bytes that must cross HBM once (Q rows that exist, edge stream, rows written) vs 8 TB/s
        #   l2_*                     one D-wide Q row per edge, served by L2 / MALL, vs the L2 bandwidth of the guide
        kname = ("k_mpnn_win (window form: distinct source rows of a window of targets staged in LDS, MFMA mat-vec; "
                 f"{win} of {n} launches; `l2_*` quotes the per-edge gather volume it no longer moves)" if win * 2 > n else
                 "k_mpnn_max / k_mpnn_fast (fused gather + per-edge mat-vec + segmented reduce)")
        gather = {"bound": "hbm", "kernel": kname,
                  "achieved": hbm if hbm is not None else comp, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": (hbm if hbm is not None else comp) / PEAK_HBM_GBS,
                  "traffic": traffic, "traffic_over_compulsory": (traffic * n / agg["bytes"]) if traffic else None,
                  "traffic_source": (pmc.get("live") or "profiles/pmc_mpnn_summary.json (rocprofv3 --pmc passes of an earlier run of this "
                                     "command; not re-measured in this run)") if traffic else None,
                  "compulsory_bytes_per_launch": agg["bytes"] / n, "compulsory_gbs": comp, "compulsory_frac": comp / PEAK_HBM_GBS,
                  "l2_gather_bytes_per_launch": agg.get("gather_bytes", 0.0) / n, "l2_gbs": l2, "l2_peak": PEAK_L2_GBS,
                  "l2_frac": l2 / PEAK_L2_GBS,
                  "avg_launch_ms": agg["ms"] / n, "valu_tflops": agg["flops"] / sec / 1e12,
                  "share_of_step_ms": agg["ms"] / steps}
    return roofline, gather


def instrumented(model, settings, batches, steps, symmetric):
    """`steps` eager passes over `batches` with HIP events inside librgnn around the two dominant kernels (rank 0)."""
    from radargnn_amd import frames as fr, ops
    from radargnn_amd.gnn import mpnn_layers
    eager = fr.HotPath(model, settings, use_hip_graphs=False)
    _, _, g = eager(batches[0])
    rows = int(torch.unique(g.edge_index[1]).numel()) if g.edge_index.numel() else 0
    prof = EventProfiler(rows_with_edges=rows, symmetric=symmetric)
    ops.ctx().profiler = prof
    prof.enabled = True
    side, plan_side = mpnn_layers.ISO_SIDE_STREAM, mpnn_layers.PLAN_ON_SIDE_STREAM
    mpnn_layers.ISO_SIDE_STREAM = False          # every launch alone on the device while it is being timed
    mpnn_layers.PLAN_ON_SIDE_STREAM = False      # (the window plan's kernels otherwise run beside the embedding / first dense launches)
    try:
        for _ in range(steps):
            for b in batches:
                eager(b)
        torch.cuda.synchronize()
    finally:
        mpnn_layers.ISO_SIDE_STREAM, mpnn_layers.PLAN_ON_SIDE_STREAM = side, plan_side
        prof.enabled = False
        ops.ctx().profiler = None
    return prof.summary()


def search_roofline(batch, settings, n_edges, reps=20):
    """The graph-construction stage on its own (grid build, search count / scan / fill, row sort, features, degree; the CSR
    by target belongs to the model stage): HIP-event time per batch against the bytes that must cross HBM at least once --
    48 B per point (positions, velocities, rcs, timestamp in, node features out) + 16 B per edge (edge_index) + the edge
    attributes.  It is a latency-bound chain of small kernels, nowhere near a bandwidth roofline; the number says how far."""
    from radargnn_amd import frames as fr
    for _ in range(3):
        g = fr.build_graphs(batch, settings)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g = fr.build_graphs(batch, settings)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    n = int(batch.num_points)
    bytes_min = 48 * n + 16 * n_edges + 4 * int(g.edge_attr.shape[1]) * n_edges
    gbs = bytes_min / ms / 1e6
    return {"bound": "hbm", "kernel": "graph construction stage (k_grid_frame, k_radius count, scan, k_radius_rows, node features: 7 launches), one host read of the edge count included",
            "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None,
            "bytes_per_batch": bytes_min, "ms_per_batch": ms,
            "note": "latency-bound chain of 7 launches + one host read (eager); bandwidth is not what limits it"}


def other_config(name, model, settings, frame_batches, steps, unit_frames, symmetric, roofs=True, bn_scope="batch"):
    """One of the other BASELINE.json configurations after the timed region: wall time of `steps` passes over its
    resident batches (eager launches), then an instrumented pass for the roofline fraction of its dominant kernel."""
    from radargnn_amd import frames as fr
    model = model.cuda()
    batches = [fr.FrameBatch.from_frames(fb) for fb in frame_batches]
    hot = fr.HotPath(model, settings, use_hip_graphs=False, bn_scope=bn_scope)
    for b in batches[:2] * 2:
        _, _, g = hot(b)
    g.check()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for b in batches:
            hot(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mode, probe = "eager launches", {"eager": dt * 1e3}
    if len(batches) == 1:
        # a single resident batch: the HIP-graph replay of everything behind the search is the faster launch mode when the step
        # is launch-bound (one small frame); report the faster one, like the headline does
        hg = fr.HotPath(model, settings, use_hip_graphs=True, bn_scope=bn_scope)
        for _ in range(4):
            hg(batches[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            hg(batches[0])
        torch.cuda.synchronize()
        dg = (time.perf_counter() - t0) / steps
        probe["graph"] = dg * 1e3
        if dg < dt:
            dt, mode = dg, "one HIP graph per step (search, features, CSR build and model), replayed"
        del hg
    roof = gather = None
    if roofs:
        summ = instrumented(model, settings, batches[:1], 2, symmetric)
        roof, gather = rooflines(summ, 2, with_pmc=False)
    cands = [r for r in (roof, gather) if r]
    dom = max(cands, key=lambda r: r["share_of_step_ms"]) if cands else None
    n_frames = sum(len(fb) for fb in frame_batches)
    out = {"config": name, "batches": len(batches), "frames": n_frames, "points": int(sum(b.num_points for b in batches)),
           "edges_first_batch": int(g.edge_index.shape[1]), "ms_per_batch": dt / len(batches) * 1e3,
           "ms_per_pass": dt * 1e3, "frames_per_s": n_frames / dt, "unit": unit_frames, "launch_mode": mode,
           "launch_mode_probe_ms": probe}
    if dom:
        out["dominant_kernel"] = dom["kernel"].split(" ")[0]
        out["dominant_bound"] = dom["bound"]
        out["dominant_frac"] = dom.get("compulsory_frac", dom["frac"])      # (edge kernel: on its compulsory bytes)
        out["dominant_ms_per_batch"] = dom["share_of_step_ms"]
        if "l2_frac" in dom:
            out["dominant_l2_frac"] = dom["l2_frac"]
        if "mfma" in dom:
            out["dominant_mfma_frac"] = dom["mfma"]["frac"]
    return out


def training_step(steps=4):
    """One training step of the C2 workload the way the reference's trainer drives it (gnn/trainer.py:175-231: zero_grad,
    requires_grad_ on the inputs, forward, cross-entropy + Huber loss, backward, Adam) -- SURVEY 8(f) row 1; measured after the
    timed region, not part of the headline."""
    from radargnn_amd import frames as fr, synthetic
    from radargnn_amd.gnn.losses import detection_loss
    model = c2_model().cuda()
    batch = fr.FrameBatch.from_frames([synthetic.radarscenes_frame(i) for i in range(FRAMES_PER_GPU)])
    g = fr.build_graphs(batch, c2_settings())
    n = g.x.shape[0]
    gen = torch.Generator(device="cuda").manual_seed(0)
    y = torch.cat((torch.randint(0, 6, (n, 1), device="cuda", generator=gen).float(), torch.randn(n, 5, device="cuda", generator=gen)), 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, ei, ea = g.x, g.edge_index, g.edge_attr

    def step():
        opt.zero_grad()
        x.requires_grad_(); ea.requires_grad_()
        c, bb = model(x, ei, ea)
        loss, _, _ = detection_loss(c, bb, y, 5, [1.0, 1.0, 1.0, 1.0, 1.0, 0.3])
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"config": "training step on the C2 workload (64 frames x 3000 pts, radius graph built once; forward + loss + backward + Adam, "
                      "the model's public forward: no symmetry / visiting-order hints)",
            "ms_per_step": dt * 1e3, "frames_per_s": FRAMES_PER_GPU / dt, "loss": float(loss.item())}


def other_configs():
    """C1, C3, one rank's share of C4 (the real loop: 1024 frames in 16 batches of 64) and C5 -- SURVEY 8(d) shapes."""
    from radargnn_amd import frames as fr, synthetic
    out = []
    rs = lambda a, b: [synthetic.radarscenes_frame(i) for i in range(a, b)]
    out.append(other_config("C1: 1 frame x 3000 pts, kNN k=10, 2-layer MPNNConv [224,224] (latency case)",
                            shipped_model([224, 224], 6), fr.GraphSettings(algorithm="knn", k=10), [rs(0, 1)], 50,
                            "frames", False))
    out.append(other_config("C2 with PER-FRAME BatchNorm statistics (bn_scope='frame'): the numbers of the reference's one-frame-per-"
                            "forward inference loop (evaluate.py:40, model never in eval mode) at batched throughput",
                            c2_model(), c2_settings(), [rs(0, FRAMES_PER_GPU)], 10, "frames", True, roofs=False, bn_scope="frame"))
    out.append(other_config("C3: 512 nuScenes-shaped frames x 300 pts, kNN k=20, shipped 5-layer model, 11 classes",
                            shipped_model([224, 224, 128, 64, 32], 11), fr.GraphSettings(algorithm="knn", k=20),
                            [[synthetic.nuscenes_frame(i) for i in range(512)]], 10, "frames", False))
    out.append(other_config("C4, one rank's share: 1024 RadarScenes-shaped frames in 16 batches of 64, kNN k=20, shipped "
                            "5-layer model + both heads (8 ranks run this independently, no collective)",
                            shipped_model([224, 224, 128, 64, 32], 6), fr.GraphSettings(algorithm="knn", k=20),
                            [rs(64 * b, 64 * (b + 1)) for b in range(16)], 2, "frames", False))
    out.append(other_config("C5: one 100k-point cloud, radius r=1, 6-layer model on rotation-invariant features",
                            shipped_model([224, 224, 224, 128, 64, 32], 6, node_dim=4, edge_dim=4),
                            fr.GraphSettings(algorithm="radius", r=1.0,
                                             node_features=("rcs", "velocity_vector_length", "time_index", "degree"),
                                             edge_features=("point_pair_features",)), [[synthetic.stress_cloud()]], 10,
                            "clouds", True))
    return out


def single_frames_streamed(n_frames=256):
    """The reference's own inference regime (evaluate.py:40 ``batch_size=1``, postprocessor/inference.py:48-68): ONE frame per
    forward, host frame in, host logits / boxes out, train-mode BatchNorm over that frame's nodes.  Every frame is a different graph,
    so the launches are eager (~75 ctypes launches per frame): this number is bound by the launching thread, not by the device.
    Beside it: the same frames as one 64-frame batch with per-frame statistics (bn_scope='frame') -- what `other_configs` reports as
    batched throughput -- must give the same numbers, checked here on the first 64 frames."""
    from radargnn_amd import frames as fr, synthetic
    model, settings = c2_model().cuda(), c2_settings()
    frames = [synthetic.radarscenes_frame(i) for i in range(FRAMES_PER_GPU)]
    streamer = fr.FrameStreamer(fr.HotPath(model, settings, use_hip_graphs=False))
    warm = max(4 * streamer.slots, 24)

    def batches():
        for i in range(warm + n_frames):
            yield [frames[i % len(frames)]]

    stamps, kept = [], []
    for i, (cls, bb) in enumerate(streamer.run(batches())):
        stamps.append(time.perf_counter())
        if warm <= i < warm + len(frames):
            kept.append((cls.clone(), bb.clone()))
    timed = stamps[warm - 1:]
    gaps = sorted(b - a for a, b in zip(timed[:-1], timed[1:]))
    batch = fr.FrameBatch.from_frames(frames)
    b_cls, b_bb, _ = fr.HotPath(model, settings, bn_scope="frame")(batch)
    b_cls, b_bb = b_cls.cpu(), b_bb.cpu()
    ptr = batch.frame_ptr.cpu().tolist()
    err_c = err_b = 0.0
    for k, (c, b) in enumerate(kept):
        f = (warm + k) % len(frames)
        rc, rb = b_cls[ptr[f]:ptr[f + 1]], b_bb[ptr[f]:ptr[f + 1]]
        err_c = max(err_c, float((c - rc).abs().max() / rc.abs().max()))
        err_b = max(err_b, float((b - rb).abs().max() / rb.abs().max()))
    return {"config": "single frames streamed, batch_size = 1: C2's model and radius graph, one 3000-point frame per forward, host in / host out "
                      "(the reference's inference loop: evaluate.py:40, inference.py:48-68)",
            "frames": n_frames, "value": n_frames / (timed[-1] - timed[0]), "unit": "frames/s",
            "ms_per_frame": (timed[-1] - timed[0]) / n_frames * 1e3, "median_interval_ms": gaps[len(gaps) // 2] * 1e3,
            "bound_by": "the launching thread (eager launches: every frame is a different graph)",
            "vs_the_same_frames_batched_with_bn_scope_frame": {"logits": err_c, "boxes": err_b, "frames_compared": len(kept),
                                                               "note": "norm-wise max |a - b| / max |b| per frame, worst frame"}}


def pin_rank_to_cores(local_rank, local_world):
    """One process per GPU on one host: every rank keeps to its own block of the cores this process may run on (the launching
    thread enqueues ~75 launches per streamed batch, the streamer adds a loader thread: eight ranks left to the scheduler migrate
    and contend).  Blocks are contiguous slices of the allowed set, which on the two-socket MI355X hosts keeps a rank on one
    socket; RGNN_BENCH_NO_AFFINITY=1 leaves the affinity alone.  Returns what was set (for the bench line) or None."""
    if local_world <= 1 or os.environ.get("RGNN_BENCH_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, torch.get_num_threads())))
        return {"cores": [mine[0], mine[-1]], "count": len(mine)}
    except OSError:
        return None


def self_launch_command(gpus, env, argv, script=None, port=None):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment) starts its own N ranks: the
    command that re-runs this script under torch.distributed.run, one process per GPU of this node, or None when the process
    is already a rank (WORLD_SIZE set -- the driver's own torchrun line) or a single GPU was asked for.  No reference
    counterpart: the reference is single-device (gnn/trainer.py:65, postprocessor/inference.py:43)."""
    if "WORLD_SIZE" in env or (gpus <= 1 and not env.get("RGNN_BENCH_SELF_LAUNCH")):   # (the switch: the launcher path on 1 GPU)
        return None
    script = script or os.path.abspath(__file__)
    port = port or env.get("MASTER_PORT") or str(29500 + (os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script, *argv]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-streaming measurement (loader thread + three streams: rocprofv3 --pmc "
                                                           "serialises dispatches and never finishes it)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not start the two rocprofv3 --pmc passes that measure roofline.traffic / "
                    "step_traffic in this run (live_traffic); the line then quotes the committed summaries under profiles/")
    ap.add_argument("--no-c4", action="store_true", help="under a process group: skip the C4 block (every rank's 1024-frame share)")
    ap.add_argument("--launch-mode", choices=["auto", "eager", "graph"], default="auto",
                    help="eager: plain launches on one stream; graph: the post-search launches are replayed from one "
                         "captured HIP graph (same kernels, same order); auto (default): both are timed for a few untimed "
                         "steps and the faster one runs the timed region -- on an idle host both are GPU-bound, on a loaded "
                         "host the eager enqueue (about 75 ctypes launches per step) becomes the bottleneck")
    ap.add_argument("--hip-graphs", action="store_true", help="same as --launch-mode graph")
    ap.add_argument("--cpu-frames", type=int, default=4)
    a = ap.parse_args()

    cmd = self_launch_command(a.gpus, os.environ, sys.argv[1:])
    if cmd is not None:                                          # --gpus N without a launcher: start the N ranks ourselves
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # (dmabuf IPC: what RCCL needs on this driver)
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:                                          # never print a line whose n_gpus is not what was asked for
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    affinity = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    dist = None
    if world > 1 or os.environ.get("RGNN_BENCH_FORCE_DIST"):   # the env switch exercises the RCCL code path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from radargnn_amd import frames as fr
    from radargnn_amd import synthetic

    settings = c2_settings()
    model = c2_model().cuda()                                    # training mode on purpose (reference behaviour)
    if a.hip_graphs:
        a.launch_mode = "graph"
    first, last = rank * FRAMES_PER_GPU, (rank + 1) * FRAMES_PER_GPU          # weak scaling: own frames per rank
    frames_list = [synthetic.radarscenes_frame(i) for i in range(first, last)]
    batch = fr.FrameBatch.from_frames(frames_list)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def probe(h, steps=6):
        for _ in range(3):                                       # eager warm-up, graph capture, first replay
            h(batch)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            h(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps

    probes = {}
    if a.launch_mode == "auto":                                  # untimed: pick the launch mode this host is faster with
        cand = {"eager": fr.HotPath(model, settings, use_hip_graphs=False),
                "graph": fr.HotPath(model, settings, use_hip_graphs=True)}
        probes = {k: probe(h) for k, h in cand.items()}
        use_graph = probes["graph"] < 0.99 * probes["eager"]     # (C2: the replay is 1 - 1.5 % faster than eager launches)
        if dist is not None:                                     # all ranks run the same mode (rank 0 decides)
            flag = torch.tensor([1 if use_graph else 0], device="cuda")
            dist.broadcast(flag, 0)
            use_graph = bool(flag.item())
        hot = cand["graph" if use_graph else "eager"]
        del cand
    else:
        use_graph = a.launch_mode == "graph"
        hot = fr.HotPath(model, settings, use_hip_graphs=use_graph)
    a.hip_graphs = use_graph
    for _ in range(max(a.warmup, 3 if use_graph else 0)):        # >= 3 passes: eager warm-up, graph capture, first replay
        cls, bb, g = hot(batch)
    g.check()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cls, bb, g = hot(batch)
    sync_all()
    elapsed = own_elapsed = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)
        per_rank = [FRAMES_PER_GPU * a.steps / float(x.item()) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- self check (every rank, after the timed region): what the timed steps left in the output buffers is finite, the
    # device status words are clean, and it equals ONE eager pass over the same batch bit for bit (same kernels, same order;
    # train-mode logits do not depend on the running statistics the steps kept updating).  A step that returned early or
    # replayed a stale graph cannot print a number.
    t_cls, t_bb = cls.clone(), bb.clone()
    g.check()
    e_cls, e_bb, e_g = fr.HotPath(model, settings, use_hip_graphs=False)(batch)
    e_g.check()
    if not (bool(torch.isfinite(t_cls).all()) and bool(torch.isfinite(t_bb).all())):
        raise SystemExit("bench.py self check FAILED: non-finite logits / boxes after the timed region")
    if not (torch.equal(t_cls, e_cls) and torch.equal(t_bb, e_bb) and torch.equal(g.edge_index, e_g.edge_index)):
        raise SystemExit("bench.py self check FAILED: the timed steps' outputs differ from one eager pass over the same batch")
    self_check = (f"ok: outputs of the timed steps finite and bit-equal to one eager pass ({t_cls.shape[0]} x {t_cls.shape[1]} "
                  f"logits, {t_bb.shape[0]} x {t_bb.shape[1]} boxes, {int(g.edge_index.shape[1])} edges); device status clean")

    # ---- further blocks of the same timed loop, the eval-mode step and the fp32-MFMA step (rank 0 of a single-GPU run; all after
    # the timed region and the self check): a 2 % change of a kernel is invisible in ONE 20-step block on a pool whose boxes differ
    # by 4 %, the spread of 15 blocks in one process says what a block of this box is worth
    repeats = eval_mode = fp32_line = None
    if rank == 0 and world == 1 and not a.no_other_configs:      # (the profiling scripts pass --no-other-configs: their kernel statistics stay those of the timed loop)
        def block(h, steps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(steps):
                h(batch)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / steps * 1e3
        blocks = sorted(block(hot, a.steps) for _ in range(15))
        repeats = {"n": len(blocks), "steps_per_block": a.steps, "ms_per_step": {"median": blocks[len(blocks) // 2], "min": blocks[0], "max": blocks[-1]},
                   "median": FRAMES_PER_GPU / (blocks[len(blocks) // 2] * 1e-3), "min": FRAMES_PER_GPU / (blocks[-1] * 1e-3),
                   "max": FRAMES_PER_GPU / (blocks[0] * 1e-3), "unit": "frames/s",
                   "note": "further blocks of the timed loop in the same process, after the block `value` reports"}

        def variant(prepare, restore):
            prepare()
            try:
                h = fr.HotPath(model, settings, use_hip_graphs=use_graph)
                for _ in range(3):
                    h(batch)[2].check()
                ms = sorted(block(h, a.steps) for _ in range(3))[1]
                return {"ms_per_step": ms, "value": FRAMES_PER_GPU / (ms * 1e-3), "unit": "frames/s"}
            except Exception as e:                                   # (a line that cannot be measured says why)
                return {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                restore()
        # "also report eval-mode numbers separately" -- BatchNorm applies its running statistics: no column statistics
        # in the dense epilogues, no finalize launches
        eval_mode = variant(model.eval, model.train)
        if isinstance(eval_mode, dict) and "value" in eval_mode:
            eval_mode["note"] = "model.eval(): BatchNorm from the running statistics (the reference's evaluate.py never calls eval(); SURVEY 8d asks for the number)"
        from radargnn_amd import ops as _ops

        def fp32_on():
            os.environ["RGNN_LINEAR_FP32"] = "1"
            _ops.reload_env()

        def fp32_off():
            os.environ.pop("RGNN_LINEAR_FP32", None)
            _ops.reload_env()
        fp32_line = variant(fp32_on, fp32_off)
        if isinstance(fp32_line, dict) and "value" in fp32_line:
            fp32_line["note"] = ("RGNN_LINEAR_FP32=1: every dense layer on v_mfma_f32_32x32x2_f32 (true fp32 operands) -- what the f16x2 form "
                                 "(2 f16 terms, 3 products) is chosen over, measured on this box in this run")
            fp32_line["step_time_over_the_f16x2_step"] = fp32_line["ms_per_step"] / repeats["ms_per_step"]["median"]

    # ---- C4 under a process group: every rank runs ITS share of BASELINE.json configs[3] (8 x 1024 frames: 16 resident
    # batches of 64, kNN k = 20, shipped 5-layer model + both heads; no collective in the data path), rank 0 reports the sum
    c4 = None
    if dist is not None and not a.no_c4:
        per = 16 * FRAMES_PER_GPU
        mine = other_config("C4 share of this rank", shipped_model([224, 224, 128, 64, 32], 6), fr.GraphSettings(algorithm="knn", k=20),
                            [[synthetic.radarscenes_frame(rank * per + 64 * b + i) for i in range(64)] for b in range(16)],
                            2, "frames", False, roofs=(rank == 0))
        sync_all()
        t = torch.tensor([mine["ms_per_pass"]], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)
        ms = [float(x.item()) for x in every]
        c4 = {"config": "C4: %d x 1024 RadarScenes-shaped frames, frame-sharded (16 resident batches of 64 per rank), kNN k=20, "
                        "shipped 5-layer model + both heads; no collective in the data path" % world,
              "frames_per_s_total": world * per / (max(ms) * 1e-3), "per_rank_frames_per_s": [per / (m * 1e-3) for m in ms],
              "ms_per_batch": max(ms) / 16, "ms_per_pass_max_over_ranks": max(ms), "rank0": mine}

    if rank == 0:
        # instrumented pass: the same steps launched eagerly with HIP events recorded inside librgnn around the two dominant
        # kernels -- HIP graph replay cannot carry timing events.  Same kernels, same shapes, same stream.
        summ = instrumented(model, settings, [batch], a.steps, symmetric=True)
        live_note = None
        under_profiler = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
        if world == 1 and not a.no_live_traffic and not under_profiler:
            torch.cuda.synchronize()
            live_note = live_traffic()                      # (None: measured; else why not -- the line says so)
        roofline, gather = rooflines(summ, a.steps)
        line = {
            "metric": "radar frames/sec (graph-build + GNN fwd)",
            "value": world * FRAMES_PER_GPU * a.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (dense layers: fp32 operands split into 2 f16 terms after an exact power-of-two pre-scale, 3 f16 MFMA "
                     "products per fp32 product, fp32 accumulate -- more accurate than fp32 MFMA against float64, "
                     "tests/test_gpu_f16x2.py; RGNN_NO_F16X2=1 selects 3 bf16 terms / 6 products, RGNN_LINEAR_FP32=1 "
                     "v_mfma_f32_32x32x2_f32)",
            "data": "synthetic",
            "config": {"workload": "C2: per GPU 64 RadarScenes-shaped frames x 3000 pts, radius graph r=1.0, node feats "
                                   "[rcs,velocity_vector,time_index,degree], edge feats [relative_position], 4-layer "
                                   "MPNNConv [224,224,128,64] + emb MLPs + both heads, train-mode BatchNorm, max aggr",
                       "launch_mode": ("one HIP graph per step (search, features, CSR build and model; edge count verified on the device), replayed"
                                       if a.hip_graphs else "eager launches on one stream (GPU-bound: launch queue stays ahead), 1 host read of E"),
                       "launch_mode_probe_ms": {k: round(v * 1e3, 3) for k, v in probes.items()} or None,
                       "frames_per_gpu": FRAMES_PER_GPU, "points_per_gpu": int(batch.num_points),
                       "edges_per_gpu": int(g.edge_index.shape[1]), "sharding": "frames, no collective",
                       "ranks_in_process_group": (dist.get_world_size() if dist is not None else 1),
                       "per_rank_frames_per_s": per_rank},
            "roofline": roofline,
            "self_check": self_check,
        }
        if repeats is not None:
            line["value_repeats"] = repeats
            line["eval_mode"] = eval_mode
            line["fp32_mfma"] = fp32_line
        if affinity is not None:
            line["config"]["cpu_affinity_rank0"] = affinity
        if c4 is not None:
            line["c4"] = c4
        if gather:
            line["roofline_gather"] = gather
        line["roofline_search"] = search_roofline(batch, settings, int(g.edge_index.shape[1]))
        line["step_traffic"] = step_traffic(int(batch.num_points), int(g.edge_index.shape[1]))
        line["live_traffic"] = ("counters measured in this run (roofline.traffic, roofline_gather.traffic, step_traffic)" if LIVE_PMC else
                                f"not measured in this run ({live_note or 'switched off, more than one rank, or already under a profiler'}): "
                                "the committed summaries under profiles/ are quoted")
        line["bytes_over_survey_compulsory"] = line["step_traffic"]["bytes_over_survey_compulsory"]
        line["parity_margin"] = parity_margin()
        if not a.no_pcie:
            pc = pcie_inclusive(fr.HotPath(model, settings, use_hip_graphs=False), frames_list, max(200, 2 * a.steps))
            line["pcie_inclusive_value"] = pc["value"]                    # whole window, stalls included
            line["pcie_inclusive"] = {"batches": pc["batches"], "from_the_median_batch_interval": pc["median_interval_value"],
                                      "over_resident_value": pc["value"] / line["value"],
                                      "median_over_resident_value": pc["median_interval_value"] / line["value"],
                                      "longest_interval_ms": pc["longest_interval_ms"], "slots": pc["slots"],
                                      "results_behind": pc["results_behind"]}
        if world == 1 and not a.no_other_configs:
            line["other_configs"] = other_configs()
            line["training_step"] = training_step()
            try:
                line["other_configs"].append(single_frames_streamed())
            except Exception as e:                                   # (a side measurement must not cost the line)
                line["other_configs"].append({"config": "single frames streamed, batch_size = 1", "error": f"{type(e).__name__}: {e}"[:300]})
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, settings, a.cpu_frames)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    else:
        line = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # RCCL prints a version banner through C stdio; flush it (and anything else buffered in libc) BEFORE the result so
        # that the JSON line is the last line on stdout, then leave without running further native teardown prints
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
