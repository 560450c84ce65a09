#!/usr/bin/env python3
"""GPU diagnostic (needs the -DDSS2_STAMPS build, DSS2_LIB=.../libdss2_hip_stamps.so): per-wave phase
durations of gemm_prop (H->H forward, C2) from in-kernel s_memtime stamps."""
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
n = topo.ntiles * 4 * 8
buf = (C.c_ulonglong * n)()
assert lib.dss2_debug_read_stamps(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)   # s_memtime: 100 MHz constant clock? (ticks)
names = ["staging+barrier (1->2)", "MFMA loop (2->3)", "Horner (3->4)", "stores (4->5)", "tile total (1->5)"]
d = [t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4], t[:, 5] - t[:, 1]]
t0 = t[:, 1].min()
print(f"tiles={topo.ntiles} waves={t.shape[0]} kernel span (first tile start -> last store) {t[:, 5].max() - t0} ticks")
for nm, v in zip(names, d):
    print(f"{nm:24s} mean {v.mean():9.0f}  median {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f} ticks  ({100 * v.mean() / d[4].mean():5.1f}% of tile)")
g = t.reshape(topo.ntiles, 4, 8)
nwg = min(topo.ntiles, 512)
first, second = g[:nwg], g[nwg:2 * nwg]
print("tile-1 start (median over WGs, rel. kernel start): round-1 WGs", int(np.median(first[:256, 0, 1]) - t0), " round-2 WGs", int(np.median(first[256:, 0, 1]) - t0))
if len(second):
    print("gap tile-1 end -> tile-2 start (barrier):", int(np.median(second[:, 0, 1] - first[:len(second), :, 5].max(axis=1))),
          "ticks; tile-2 staging:", int(np.median(second[:, 0, 2] - second[:, 0, 1])), " tile-1 staging:", int(np.median(first[:, 0, 2] - first[:, 0, 1])))
    print("tile-1 total:", int(np.median(first[:, :, 5].max(axis=1) - first[:, 0, 1])), " tile-2 total:", int(np.median(second[:, :, 5].max(axis=1) - second[:, 0, 1])))
