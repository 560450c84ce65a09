#!/usr/bin/env python3
"""GPU diagnostic (needs a -DDSS2_STAMPS build of csrc/dss2_wgrad16q.hip: DSS2_LIB=<that library>): per-wave slot durations of
wgrad16q_kernel on the third tile of every workgroup's range (s_memtime ticks), C2."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw, ops = pkg.networks, pkg.ops
DEV = "cuda:0"; H, nmat, nl = 128, 3, 3
b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=0)
ei = b["edge_index"].to(DEV); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
torch.manual_seed(0)
Ws = [[torch.randn(H, H, device=DEV) * (1.2 / H ** 0.5) for _ in range(nmat)] for _ in range(nl)]
plan = nw._PackPlan(Ws, DEV, bf16_groups=tuple(range(nl))); plan.refresh()
h = torch.randn(N, H, device=DEV)
xps = [ops.new_xplanes(topo, H, DEV) for _ in range(nl)]
ops.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[i], Y=torch.empty(N, H, device=DEV), relu=True, x_planes=xps[i]) for i in range(nl)], b_format=1)
Gs = [torch.randn(N, H, device=DEV) for _ in range(nl)]
stride = nmat * H * H + H
out = torch.empty(nl * stride, device=DEV)
for _ in range(5):
    ops.wgrad_batched_xp(topo, Gs, H, xps, H, nmat, out, pending=[])
torch.cuda.synchronize()
lib = C.CDLL(pkg._lib.LIB_PATH)
n = 256 * 8 * 16
buf = (C.c_ulonglong * n)()
assert lib.dss2_debug_read_qstamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
def show(name, v):
    print(f"{name:62s} mean {v.mean():8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f} ticks")
for role, sl in (("waves 0-3 (G, P G planes)", slice(0, 4)), ("waves 4-7 (P^2 G planes)", slice(4, 8))):
    print(role)
    tt = t[:, sl, :]
    show("  slot A : 36 MFMAs || planes of chunk 1, bias sums", tt[:, :, 1] - tt[:, :, 0])
    show("  barrier", tt[:, :, 2] - tt[:, :, 1])
    show("  slot B0:  4 MFMAs || next tile's rows -> LDS", tt[:, :, 3] - tt[:, :, 2])
    show("  barrier", tt[:, :, 4] - tt[:, :, 3])
    show("  slot B1: 12 MFMAs || first hop of the next tile", tt[:, :, 5] - tt[:, :, 4])
    show("  barrier", tt[:, :, 6] - tt[:, :, 5])
    show("  slot B2: 20 MFMAs || planes of the next tile's chunk 0", tt[:, :, 7] - tt[:, :, 6])
    show("  barrier", tt[:, :, 8] - tt[:, :, 7])
    show("  tile total (72 MFMAs = 2304 cycles of matrix pipe per wave)", tt[:, :, 8] - tt[:, :, 0])
