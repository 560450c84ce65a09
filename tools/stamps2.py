#!/usr/bin/env python3
"""GPU diagnostic (-DDSS2_STAMPS build): per-step work / barrier-wait durations of the two-team kernel."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
dev = torch.device("cuda:0"); H, nmat = 128, 3
b = pkg.synthetic.make_batch(["cigre14"], 4096, seed=0)
ei = b["edge_index"].to(dev); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
plan = nw._PackPlan([Ws], dev); plan.refresh()
h = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev); bias = torch.randn(H, device=dev)
for _ in range(5):
    nw.gemm_prop(topo, h, H, H, plan.fwd[0], nmat, H, out, bias=bias, relu=True)
torch.cuda.synchronize()
lib = C.CDLL(pkg._lib.LIB_PATH)
n = 256 * 8 * 16
buf = (C.c_ulonglong * n)()
assert lib.dss2_debug_read_stamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 16).astype(np.int64)
for s in range(5):
    w0, w1, w2 = t[:, :, 3 * s], t[:, :, 3 * s + 1], t[:, :, 3 * s + 2]
    for team in (0, 1):
        sl = slice(0, 4) if team == 0 else slice(4, 8)
        role = "MFMA" if (s & 1) == team else ("epilogue+stage" if s >= 1 else "idle")
        work = (w1 - w0)[:, sl]; wait = (w2 - w1)[:, sl]
        print(f"step {s} team {team} {role:15s} work mean {work.mean():8.0f} p90 {np.percentile(work, 90):8.0f}   barrier wait mean {wait.mean():8.0f}")
tot = t[:, :, 14] - t[:, :, 0]
print("whole loop per wave: mean", int(tot.mean()), "max", int(tot.max()))
