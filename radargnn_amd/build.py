"""Compile radargnn_amd/csrc/*.hip for gfx950 into radargnn_amd/librgnn.so (in-tree, so it travels with the repo
snapshot to the GPU box).  hipcc cross-compiles without a GPU.

    python -m radargnn_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "librgnn.so")
SOURCES = ["core.hip", "graph.hip", "features.hip", "linear.hip", "linear_dma.hip", "mpnn.hip", "mpnn_tiles.hip", "norm.hip", "backward.hip", "wgrad.hip", "collate.hip", "postprocess.hip", "loss.hip", "embed.hip"]
# per-file flags.  mpnn_tiles.hip: its running maxima take MFMA results; without -fno-honor-nans hipcc quiets every operand of
# every fmaxf with a `v_max_f32 x, x, x` of its own (twice the vector instructions of the segmented maximum)
EXTRA_FLAGS = {"mpnn_tiles.hip": ["-fno-honor-nans"]}
# -ffp-contract=off: the neighbour search must not fuse multiply-adds (bit-exact float64 distances, see
# graph.hip); kernels that want FMAs ask for them explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
         # MFMA accumulators stay in the unified VGPR file: without this hipcc copies all 64 accumulator registers
         # between VGPRs and AGPRs around every k-step of the dense kernel (measured +15 % on the GEMMs)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _deps(src: str):
    return [src, os.path.join(CSRC, "common.h"), os.path.join(CSRC, "linear_common.h"), os.path.join(CSRC, "stats.h"),
            os.path.join(os.path.dirname(HERE), "include", "rgnn.h")]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name.replace(".hip", ".o"))
        if force or _stale(obj, _deps(src)):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print("[radargnn_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, n.replace(".hip", ".o")) for n in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print("[radargnn_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
