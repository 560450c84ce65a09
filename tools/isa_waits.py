#!/usr/bin/env python3
"""Scan the gfx950 code objects of csrc/obj/*.o for the pattern that cost the weight-gradient kernels their prefetch (round 5): a vector
memory load followed within a few instructions by an s_waitcnt vmcnt(N) with N smaller than the loads issued since -- the compiler joining
a conditional load with its alternative, a write-after-write on a register handed out while a load was in flight, two call sites merged
through temporaries.  Prints per kernel: loads, "tight" waits (<= WINDOW instructions behind a load, waiting for it), and the lines.
usage: tools/isa_waits.py [kernel-name-fragment ...]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "deep-statistical-solver-for-distribution-system-state-estimation_amd", "csrc", "obj")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
WINDOW = 8
want = sys.argv[1:]
for o in sorted(glob.glob(OBJ + "/*.o")):
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["cp", o, td + "/x.o"], check=True)
        subprocess.run([OBJDUMP, "--offloading", "x.o"], cwd=td, stdout=subprocess.DEVNULL, check=True)
        cos = glob.glob(td + "/x.o.*gfx950*")
        if not cos:
            continue
        txt = subprocess.run([OBJDUMP, "-d", "--demangle", cos[0]], capture_output=True, text=True).stdout
    name, body = None, []
    kernels = []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            if name:
                kernels.append((name, body))
            name, body = m.group(1), []
        elif name and "\t" in line:
            body.append(line.split("//")[0].strip())
    if name:
        kernels.append((name, body))
    for name, body in kernels:
        if want and not any(w in name for w in want):
            continue
        nload, tight = 0, []
        pos = []       # positions of the vector memory loads / stores so far (vmcnt counts both, in order)
        for i, ins in enumerate(body):
            if re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", ins):
                pos.append((i, ins))
                if "_load" in ins:
                    nload += 1
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
            if m and pos:
                n = int(m.group(1))
                if n < len(pos):
                    j, ld = pos[len(pos) - n - 1]          # the youngest operation this wait covers
                    if i - j <= WINDOW and "_load" in ld:
                        tight.append((i, ins, ld))
        if nload and tight:
            print(f"{os.path.basename(o):28s} {name[:110]:110s} loads {nload:4d}  tight waits {len(tight):3d}")
            if want:
                for i, ins, ld in tight[:40]:
                    print(f"      @{i:5d} {ins:28s} behind {ld}")
