"""usage: python tools/kernel_resources.py <file.hip> [name regex]
VGPRs / SGPRs / scratch / LDS / occupancy per kernel of one source of radargnn_amd/csrc (hipcc -Rpass-analysis remarks)."""
import os
import re
import subprocess
import sys

src = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "radargnn_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result",
       "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(?:[^\s:]+:\d+:\d+:\s+)?(.*?)\s+\[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    if not pat.search(name):
        continue
    g = lambda k: r.get(k, "?")
    print("%-58s VGPR %4s AGPR %3s SGPR %4s scratch %5s occ %2s LDS %s" % (name, g("VGPRs"), g("AGPRs"), g("SGPRs"),
          g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
