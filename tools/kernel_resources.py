#!/usr/bin/env python3
"""Compile one .hip file for gfx950 with -Rpass-analysis=kernel-resource-usage and print a table
(kernel, VGPRs, AGPRs, spills, scratch, occupancy).  Usage: tools/kernel_resources.py <file.hip>"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
import tempfile
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", f"-I{ROOT}/include",
       f"-I{os.path.dirname(os.path.abspath(src))}", "-c", src, "-o", tempfile.mktemp(suffix=".o", prefix="_kr"),
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
if os.path.basename(src) in ("dss2_gemm_chain16.hip", "dss2_gemm_chain_sp.hip", "dss2_gemm_chain_sp6.hip", "dss2_wgrad16.hip", "dss2_wgrad16h.hip", "dss2_wgrad16th.hip", "dss2_stack.hip", "dss2_edge16.hip"):      # as csrc/build.sh compiles it
    cmd += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':60s} {'VGPR':>5} {'AGPR':>5} {'vspill':>6} {'scratch':>7} {'occ':>3} {'SGPR':>5}")
for r in rows:
    print(f"{r['name'][:60]:60s} {r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('VGPRs Spill','?'):>6} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('Occupancy [waves/SIMD]','?'):>3} {r.get('SGPRs','?'):>5}")
