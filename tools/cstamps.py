#!/usr/bin/env python3
"""GPU diagnostic (needs the -DDSS2_CHAIN_STAMPS build: DSS2_OUT=tools/diag_lib/libdss2_cstamps.so DSS2_OBJ=/tmp/obj_cst bash csrc/build.sh -DDSS2_CHAIN_STAMPS; run with DSS2_LIB=tools/diag_lib/libdss2_cstamps.so): s_memtime phase stamps of the bf16x6
layer chain (forward; argv: graphs, hidden width (128), layers (3): the C2 shape by default) -- per layer: GEMM phase, barrier wait, Horner, epilogue, barrier wait, as the
median over workgroups and waves.  argv[1] = graphs in the batch (1024: one workgroup per CU; 4096: two per CU, two rounds)."""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw = pkg.networks
dev = torch.device("cuda:0"); nmat = 3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
GRID = sys.argv[4] if len(sys.argv) > 4 else "cigre14"
b = pkg.synthetic.make_batch([GRID], B, seed=0)
ei = b["edge_index"].to(dev); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
plan = nw._PackPlan([Ws], dev, bf16_groups=(0,)); plan.refresh()
h = torch.randn(N, H, device=dev); bias = torch.randn(H, device=dev)
outs = [torch.empty(N, H, device=dev) for _ in range(nl)]
run = lambda: nw.gemm_prop_chain(topo, h, H, nmat, [dict(Bp=plan.fwd16[0], Y=o, bias=bias, relu=True) for o in outs], b_format=1)
for _ in range(200): run()
torch.cuda.synchronize()
run(); torch.cuda.synchronize()
nwg = min(topo.ntiles, 2048)
buf = (C.c_ulonglong * (nwg * 8 * 64))()
lib = pkg._lib.lib()
sp = H >= 96 and os.environ.get("DSS2_CHAIN_SP", "1") != "0"      # the split-plane kernel keeps its stamps in its own translation unit
sp3 = topo.nrb == 3 and H >= 64 and os.environ.get("DSS2_CHAIN_SP", "1") != "0"
reader = lib.dss2_debug_read_cstamps_sp6 if sp3 else (lib.dss2_debug_read_cstamps_sp if sp else lib.dss2_debug_read_cstamps)      # (96-row tiles: gemm_chain_sp6_kernel<3, .>)
reader.argtypes = [C.c_void_p, C.c_int]
assert reader(buf, nwg * 8 * 64) == 0
ncg = (H + 31) // 32
nwav = min(8, ncg * (2 if (ncg <= 2 and topo.nrb != 3) else 1))      # waves per workgroup (row split for narrow layers)
st = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8, 64)[:, :nwav, :].astype(np.int64)
us = lambda d: float(np.median(d))      # s_memtime ticks = shader cycles
print(f"B={B}: {topo.ntiles} tiles; shader cycles (median over workgroups x waves)")
print(f"  first barrier wait: {us(st[:, :, 1] - st[:, :, 0]):.0f}")
tot = 0
for li in range(nl):
    s = lambda i: st[:, :, 2 + li * 6 + i]
    prev = st[:, :, 1] if li == 0 else st[:, :, 2 + (li - 1) * 6 + 4]
    gemm, bar1, horner, epi, bar2 = us(s(0) - prev), us(s(1) - s(0)), us(s(2) - s(1)), us(s(3) - s(2)), us(s(4) - s(3))
    epi_a = us(s(5) - s(2))
    print(f"  layer {li}: GEMM {gemm:7.0f}  barrier {bar1:5.0f}  Horner {horner:6.0f}  epilogue {epi:6.0f} (T -> stage {epi_a:5.0f}, rows -> HBM / X tile {epi - epi_a:5.0f})  barrier {bar2:5.0f}   sum {gemm + bar1 + horner + epi + bar2:7.0f} cycles")
if sp and not sp3:
    dt, drt = st[:, :, 2 + (nl - 1) * 6 + 4] - st[:, :, 1], st[:, :, 63] - st[:, :, 62]
    print(f"  in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz), median: {np.median(dt / np.maximum(drt, 1)) * 0.1:.2f} GHz")
if sp and not sp3:
    # wall-clock picture (100 MHz s_memrealtime): when workgroups start their first layer and when they end
    t0 = st[:, :, 62].min(); a = (st[:, :, 62].min(axis=1) - t0) / 100.0; e = (st[:, :, 63].max(axis=1) - t0) / 100.0
    q = lambda v: " ".join(f"{x:6.1f}" for x in np.percentile(v, [0, 10, 50, 90, 100]))
    n1 = min(nwg, 512)
    print(f"  wall clock, us after the first workgroup's start (min p10 p50 p90 max):")
    print(f"    workgroups 0..{n1 - 1}: start {q(a[:n1])} | end {q(e[:n1])} | duration {q(e[:n1] - a[:n1])}")
    if nwg > n1: print(f"    workgroups {n1}..{nwg - 1}: start {q(a[n1:])} | end {q(e[n1:])} | duration {q(e[n1:] - a[n1:])}")
wg = st[:, :, 2 + (nl - 1) * 6 + 3].max(axis=1) - st[:, :, 0].min(axis=1)
print(f"  per workgroup, first stamp -> last epilogue: median {np.median(wg):.0f} cycles, max {wg.max():.0f}")
