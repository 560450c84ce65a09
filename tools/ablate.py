#!/usr/bin/env python3
"""GPU diagnostic: time dss2_gemm_prop (TAGConv H->H forward shape, config C2) with phases switched off
through the debug bits of args.relu (1 no MFMA loop, 2 no Horner, 4 no stores, 8 no X staging)."""
import ctypes as C, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("deep-statistical-solver-for-distribution-system-state-estimation_amd")
nw, L = pkg.networks, pkg._lib
dev = torch.device("cuda:0"); H, nmat = 128, 3
GRID = sys.argv[1] if len(sys.argv) > 1 else "cigre14"; NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
b = pkg.synthetic.make_batch([GRID], NB, seed=0)
ei = b["edge_index"].to(dev); N = b["x"].shape[0]
topo = pkg.topology.get_topology(ei, N)
Ws = [torch.randn(H, H, device=dev) * 0.1 for _ in range(nmat)]
plan = nw._PackPlan([Ws], dev); plan.refresh()
h = torch.randn(N, H, device=dev); out = torch.empty(N, H, device=dev); bias = torch.randn(H, device=dev)

def args(dbg, nm=nmat, bp=None):
    a = L.GemmPropArgs()
    a.X, a.ldx, a.kreal, a.kpad = h.data_ptr(), H, H, H
    a.Bp, a.bias = (bp if bp is not None else plan.fwd[0]).data_ptr(), bias.data_ptr()
    a.Y, a.ldy, a.hout, a.ncg = out.data_ptr(), H, H, 4
    a.relu, a.nmat, a.nrb, a.ntiles = 1 | (dbg << 8), nm, topo.nrb, topo.ntiles
    a.tile_start = topo.tile_start.data_ptr()
    a.rowptr, a.col, a.w, a.max_nnz, a.ell_width = topo.rowptr.data_ptr(), topo.col.data_ptr(), topo.w.data_ptr(), topo.max_nnz, topo.ell
    a.ell_tiles = topo.ell_tiles.data_ptr()
    return a

st = torch.cuda.current_stream().cuda_stream
def t(a, reps=30):
    for _ in range(3): L.check(L.lib().dss2_gemm_prop(C.byref(a), st), "g")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().dss2_gemm_prop(C.byref(a), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(f"nrb={topo.nrb} ntiles={topo.ntiles} ell={topo.ell}")
for name, dbg in [("full", 0), ("no MFMA", 1), ("no Horner", 2), ("no stores", 4), ("no X staging", 8), ("no MFMA+Horner", 3),
                  ("only staging (no MFMA/Horner/stores)", 7), ("only MFMA (no staging/Horner/stores)", 14), ("nothing (launch+ELL+barrier)", 15),
                  ("only MFMA, no B stream", 14 | 16), ("only MFMA, no A stream", 14 | 32), ("only MFMA, no A/B streams (pure issue)", 14 | 48),
                  ("full, stagger 1", 1 << 8), ("full, stagger 2", 2 << 8), ("full, stagger 3", 3 << 8), ("full, stagger 4", 4 << 8)]:
    print(f"{name:42s} {t(args(dbg)):8.1f} us")
