#!/usr/bin/env python3
"""GPU diagnostic (needs a -DDSS2_STAMPS build of csrc/dss2_wgrad16p.hip: DSS2_LIB=<that library>): per-wave phase durations of
wgrad16p_kernel on the third tile of every workgroup's range (s_memtime ticks), C2 by default."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw, ops = pkg.networks, pkg.ops
DEV = "cuda:0"; H, nmat, nl = 128, 3, 3
GRID = sys.argv[1] if len(sys.argv) > 1 else "cigre14"; B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = pkg.synthetic.make_batch([GRID], B, seed=0)
ei = b["edge_index"].to(DEV); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
torch.manual_seed(0)
Ws = [[torch.randn(H, H, device=DEV) * (1.2 / H ** 0.5) for _ in range(nmat)] for _ in range(nl)]
plan = nw._PackPlan(Ws, DEV, bf16_groups=tuple(range(nl))); plan.refresh()
h = torch.randn(N, H, device=DEV)
Ys = [torch.empty(N, H, device=DEV) for _ in range(nl)]
xps = [ops.new_xplanes(topo, H, DEV) for _ in range(nl)]
ops.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[i], Y=Ys[i], relu=True, x_planes=xps[i]) for i in range(nl)], b_format=1)
Gs = [torch.randn(N, H, device=DEV) for _ in range(nl)]
stride = nmat * H * H + H
out = torch.empty(nl * stride, device=DEV); first = torch.empty(stride + nmat * H, device=DEV)
big = torch.empty(300 << 20, dtype=torch.uint8, device=DEV)
for _ in range(5):
    big.fill_(1)      # (cold caches, as inside the step)
    ops.wgrad_batched_xp(topo, Gs, H, xps, H, nmat, out[:(nl - 1) * stride], first_rowscale2=topo.deg_pows, first_out=first, pending=[])
torch.cuda.synchronize()
lib = C.CDLL(pkg._lib.LIB_PATH)
n = 512 * 4 * 16
buf = (C.c_ulonglong * n)()
assert lib.dss2_debug_read_pstamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 16).astype(np.int64)
def show(name, v):
    print(f"{name:58s} mean {v.mean():8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f} ticks")
show("staging (G -> LDS, ELL) + barrier", t[:, :, 1] - t[:, :, 0])
show("next item's loads issued + hop 1", t[:, :, 11] - t[:, :, 1])
show("hop-1 barrier", t[:, :, 2] - t[:, :, 11])
for c in (0, 1):
    show(f"chunk {c}: X loads issued + planes (splits, hop 2)", t[:, :, 3 + 4 * c] - (t[:, :, 2] if c == 0 else t[:, :, 6]))
    show(f"chunk {c}: barrier", t[:, :, 4 + 4 * c] - t[:, :, 3 + 4 * c])
    show(f"chunk {c}: MFMA phase (72 MFMAs, waits for X)", t[:, :, 5 + 4 * c] - t[:, :, 4 + 4 * c])
    show(f"chunk {c}: closing barrier", t[:, :, 6 + 4 * c] - t[:, :, 5 + 4 * c])
show("tile total", t[:, :, 10] - t[:, :, 0])
