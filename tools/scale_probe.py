#!/usr/bin/env python3
"""GPU: duration of the two dominant kernels vs batch size (per-launch fixed cost vs per-tile cost)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
dev = torch.device("cuda:0"); H, nmat = 128, 3
Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
plan = nw._PackPlan([Ws], dev); plan.refresh()
bias = torch.randn(H, device=dev); flat = torch.empty(nmat * H * H + H, device=dev)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# keep the chip busy first (clock ramp)
x = torch.randn(8192, 8192, device=dev)
for _ in range(50):
    x @ x
for B in (512, 1024, 2048, 3072, 4096, 6144, 8192, 16384, 32768):
    b = pkg.synthetic.make_batch(["cigre14"], B, seed=0)
    ei = b["edge_index"].to(dev); N = b["x"].shape[0]
    topo = pkg.topology.get_topology(ei, N)
    h = torch.randn(N, H, device=dev); g = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev)
    tf = timeit(lambda: nw.gemm_prop(topo, h, H, H, plan.fwd[0], nmat, H, out, bias=bias, relu=True))
    tw = timeit(lambda: nw.wgrad(topo, g, H, h, H, nmat, flat))
    print(f"B={B:6d} tiles={topo.ntiles:5d}  gemm_prop {tf:8.1f} us ({tf / topo.ntiles * 512:6.1f} us per 512 tiles)   "
          f"wgrad+reduce {tw:8.1f} us ({tw / topo.ntiles * 1024:6.1f} us per 1024 tiles)")
