#!/usr/bin/env python3
"""GPU diagnostic: s_memtime phase stamps of the whole-stack kernels (csrc/dss2_stack.hip).  Needs the stamps build:
    DSS2_OUT=tools/diag_lib/libdss2_sstamps.so DSS2_OBJ=/tmp/obj_sst bash <pkg>/csrc/build.sh -DDSS2_STACK_STAMPS
    DSS2_LIB=tools/diag_lib/libdss2_sstamps.so python tools/sstamps.py [graphs]
Prints, per phase, the median over workgroups of (max over waves) in shader cycles: forward block 0, backward last block."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
n_hh = 7
b = pkg.synthetic.make_batch(["cigre14"], B, seed=1)
x, ei, ea = b["x"].to(dev), b["edge_index"].to(dev), b["edge_attr"].to(dev)
model = pkg.SkipPFN(8, 6, 2, 32, n_hh + 1, 2, p, 5).to(dev)


def step():
    for q in model.parameters(): q.grad = None
    model(x[:, :8], ei, ea[:, :6]).square().sum().backward()


for _ in range(50): step()
torch.cuda.synchronize()
step(); torch.cuda.synchronize()
lib = pkg._lib.lib()
lib.dss2_debug_read_sstamps.argtypes = [C.c_void_p, C.c_int]
stack_mod = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd.stack")
tset = stack_mod.tiles_of(pkg.topology.get_topology(ei, x.shape[0]))
nwg = min(64, tset.ntiles)
print(f"tiles: {tset.ntiles} of {32 * tset.nrb} rows")


def read(which):
    buf = (C.c_ulonglong * (64 * 8 * 128))()
    assert lib.dss2_debug_read_sstamps(buf, which) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 128)[:nwg].astype(np.int64)


def span(st, a, b_):      # median over workgroups of (latest wave at b_) - (latest wave at a)
    return float(np.median(st[:, :, b_].max(axis=1) - st[:, :, a].max(axis=1)))


f = read(0)
print(f"B={B} p={p}: forward, block 0 (cycles, median over {nwg} workgroups)")
print(f"  staging {span(f, 0, 1):6.0f}   edge MLP {span(f, 1, 2):6.0f} (+ barrier {span(f, 2, 3):4.0f})")
for l in range(n_hh):
    s0 = 3 + 6 * l if l else 3
    base = 4 + 6 * l
    prev = 3 if l == 0 else base - 1
    print(f"  conv {l}: MFMA+mask {span(f, prev, base):6.0f}  bar {span(f, base, base + 1):4.0f}  hop1 {span(f, base + 1, base + 2):5.0f}  bar {span(f, base + 2, base + 3):4.0f}  "
          f"hop2+epilogue {span(f, base + 3, base + 4):5.0f}  bar {span(f, base + 4, base + 5):4.0f}   layer {span(f, prev, base + 5):6.0f}")
h0 = 4 + 6 * n_hh
print(f"  head {span(f, h0 - 1, h0):6.0f}    block 0 total {span(f, 1, h0):7.0f}   whole kernel (5 blocks) {span(f, 0, h0 + 1):8.0f}")
g = read(1)
print(f"backward, last block, first tile")
print(f"  staging {span(g, 0, 1):6.0f} + barrier {span(g, 1, 2):5.0f}")
for i in range(n_hh + 1):
    u = n_hh - i
    base = 3 + 5 * i
    prev = 2 if i == 0 else base - 1
    print(f"  unit {u} ({'head' if i == 0 else 'conv'}): hops {span(g, prev, base):6.0f}  MFMA phase {span(g, base, base + 1):6.0f}  bar {span(g, base + 1, base + 2):4.0f}  "
          f"gate {span(g, base + 2, base + 3):5.0f}  bar {span(g, base + 3, base + 4):4.0f}   unit {span(g, prev, base + 4):6.0f}")
    if i:      # per-role MFMA phase time
        q0 = 3 * u
        d = g[:, :, base + 1] - g[:, :, base]
        roles = {"wgrad": [(q0 + k) % 8 for k in range(3)], "dgrad": [(q0 + 3 + k) % 8 for k in range(4)], "bias sums": [(q0 + 7) % 8]}
        print("           " + "  ".join(f"{r}: {np.median(d[:, w].max(axis=1)):.0f}" for r, w in roles.items()))
e0 = 3 + 5 * (n_hh + 1)
print(f"  edge passes {span(g, e0 - 1, e0):6.0f}   dx + barrier {span(g, e0, e0 + 1):6.0f}   tile total {span(g, 0, e0 + 1):8.0f}")
